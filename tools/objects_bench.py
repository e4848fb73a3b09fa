#!/usr/bin/env python3
"""SURVEY 8f rank 3 at scale: export of a C2-sized world (1 M closed ways of 32 refs, 31 M nodes) to structure-of-arrays, the
executor's filter for all shipped match-tag sets in one call, selections -- beside the reference's own filter (apply_rule's steps
around the REAL bs_match_attr, oracle/_ref) timed on one host core over a sample.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ways", type=int, default=1_000_000)
    ap.add_argument("--refs", type=int, default=32)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="ways the host's filter is timed on, per rule")
    ap.add_argument("--cpu-rules", type=int, default=40)
    a = ap.parse_args()
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import smrender_amd as sa
    import hostsim as hs
    from test_gpu_objects import shipped_rules, bare
    sa.init(0)
    rules = shipped_rules()
    pairs = sorted({(bare(k), bare(v)) for _, _, t in rules for k, v in t if k and v and not any(c in k + v for c in "^(|")})
    pairs += [("natural", "water"), ("natural", "coastline"), ("landuse", "forest"), ("highway", "residential"), ("ele", "350"), ("ele", "1200")]
    t0 = time.perf_counter()
    nodes, ways, keep = hs.synth_world(a.ways, a.refs, pairs, 3, 50, 7)
    t_world = time.perf_counter() - t0
    t0 = time.perf_counter()
    objs = sa.Objects(nodes, ways, threads=a.threads)
    t_create = time.perf_counter() - t0
    info = objs.info
    handles, kinds = [], []
    for typ, way_kind, tags in rules:
        t = hs.OSM_NODE if typ == "node" else hs.OSM_WAY
        handles.append(hs.match_rule(t, tags, way_kind)); kinds.append(t)
    for k, v in pairs[:60]:
        handles.append(hs.match_rule(hs.OSM_WAY, [(k, v)], 0)); kinds.append(hs.OSM_WAY)
    objs.match(handles[:4])           # first call: buffers
    t0 = time.perf_counter()
    bm, counts = objs.match(handles)
    t_match = time.perf_counter() - t0
    n_way_rules = sum(k == hs.OSM_WAY for k in kinds)
    pair_count = n_way_rules * info.n_ways + (len(kinds) - n_way_rules) * info.n_nodes
    # selections of the rules with the most ways
    order = [r for r in np.argsort(-counts) if kinds[r] == hs.OSM_WAY][:10]
    t0 = time.perf_counter()
    sel_ways = 0
    for r in order:
        sel_ways += objs.select_ways(bm, int(r))[3]
    sa.sync()
    t_select = (time.perf_counter() - t0) / max(len(order), 1)
    # the reference's filter on one core, on a sample (ways only: the same rule set's way rules)
    samp = ways[:min(a.cpu_sample, len(ways))]
    way_rules = [r for r in range(len(handles)) if kinds[r] == hs.OSM_WAY][:a.cpu_rules]
    t0 = time.perf_counter()
    agree = True
    got = np.unpackbits(bm.numpy().view(np.uint8), axis=1, bitorder="little")
    t_unpack = time.perf_counter() - t0
    t0 = time.perf_counter()
    for r in way_rules:
        want = hs.filter_ptrs(handles[r], samp)
        agree = agree and bool(np.array_equal(got[r, :len(samp)], want))
    t_cpu = time.perf_counter() - t0
    cpu_pairs = len(way_rules) * len(samp)
    out = {
        "what": "objects export + rule filter", "ways": int(info.n_ways), "nodes": int(info.n_nodes), "refs": int(info.n_refs),
        "way_tags": int(info.n_way_tags), "node_tags": int(info.n_node_tags), "strings": int(info.n_strings), "pool_MB": round(info.pool_bytes / 1e6, 1),
        "create_s": round(t_create, 3), "extract_ms": round(info.extract_ms, 1), "upload_ms": round(info.upload_ms, 1), "index_ms": round(info.index_ms, 1),
        "objects_per_s_export": round((info.n_nodes + info.n_ways) / t_create),
        "rules": len(handles), "match_ms": round(t_match * 1e3, 2), "match_host_ms": round(objs.info.match_host_ms, 2), "match_device_ms": round(objs.info.match_device_ms, 2), "object_rule_pairs_per_s": round(pair_count / t_match),
        "select_ms_per_rule": round(t_select * 1e3, 3), "selected_ways": int(sel_ways),
        "cpu_reference": {"pairs_per_s": round(cpu_pairs / t_cpu), "cores": 1, "sample": f"{len(way_rules)} way rules x {len(samp)} ways", "seconds": round(t_cpu, 2),
                          "agrees_with_gpu": agree},
        "synth_world_s": round(t_world, 2),
    }
    print(json.dumps(out))
    objs.close()


if __name__ == "__main__":
    main()

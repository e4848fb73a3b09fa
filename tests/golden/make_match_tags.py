#!/usr/bin/env python3
"""Collects the MATCH TAGS of every rule of the reference's shipped rulesets and test rules (rules/**/*.osm, test/rules/*.osm:
the tags of a rule object other than `_action_`) into tests/golden/match_tags.txt, one rule per line:
    node|way <TAB> open|closed|none <TAB> key <TAB> value [<TAB> key <TAB> value ...]
(open/closed/none: whether the rule WAY's own refs close, which check_way_type, src/smrparse.c:340-351, turns into
ACTION_OPEN_WAY / ACTION_CLOSED_WAY).  Match tags are input DATA of the reference (no code); the fixture lets the GPU box check
smr_hip_objects_match against the reference's own bs_match_attr on every kind of pattern its rules really use.
Run here, where /root/reference exists."""
import glob, os
import xml.etree.ElementTree as ET
REF = "/root/reference"
out = set()
for f in sorted(glob.glob(REF + "/rules/**/*.osm", recursive=True) + glob.glob(REF + "/test/rules/*.osm")):
    try:
        root = ET.parse(f).getroot()
    except ET.ParseError:
        continue
    for el in root:
        if el.tag not in ("node", "way"):
            continue
        tags = [(t.get("k"), t.get("v")) for t in el.findall("tag")]
        if not any(k == "_action_" for k, _ in tags):
            continue
        tags = [(k, v) for k, v in tags if k != "_action_" and "\t" not in k + v and "\n" not in k + v]
        kind = "none"
        if el.tag == "way":
            nd = [n.get("ref") for n in el.findall("nd")]
            if nd:
                kind = "closed" if nd[0] == nd[-1] else "open"
        out.add("\t".join([el.tag, kind] + [x for kv in tags for x in kv]))
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "match_tags.txt")
open(p, "w").write("\n".join(sorted(out)) + "\n")
print(len(out), "distinct rules ->", p)

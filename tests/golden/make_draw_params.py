#!/usr/bin/env python3
"""Collects the parameter strings of every `draw` action in the reference's shipped rulesets and test rules
(rules/**/*.osm, test/rules/*.osm: `_action_ = draw[:params]`) into tests/golden/draw_params.txt.  Rule parameters are input
DATA of the reference (no code); the fixture lets the GPU box check that the plugin's act_draw_ini accepts every shipped rule
or refuses it for a documented reason.  Run here, where /root/reference exists."""
import glob, os, re
REF = "/root/reference"
out = set()
for f in glob.glob(REF + "/rules/**/*.osm", recursive=True) + glob.glob(REF + "/test/rules/*.osm"):
    for m in re.finditer(r"k=['\"]_action_['\"]\s+v=['\"]draw(?:@[^:'\"]*)?(?::([^'\"]*))?['\"]", open(f, errors="replace").read()):
        out.add(m.group(1) or "")
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "draw_params.txt")
open(p, "w").write("\n".join(sorted(out)) + "\n")
print(len(out), "distinct draw parameter strings ->", p)

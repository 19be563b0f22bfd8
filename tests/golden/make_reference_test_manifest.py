"""Lists the tests the reference itself holds (every `#[test]` under /root/reference, all of them in core/src/geometry) -> reference_tests.json.
Names and line numbers only: tests/test_reference_proptests.py restates each property against the oracle and must account for every entry.
Run in the build container (the reference does not travel): python3 tests/golden/make_reference_test_manifest.py"""
import json
import os
import re

REF = "/root/reference"
out = {}
for root, _, files in os.walk(REF):
    for f in sorted(files):
        if not f.endswith(".rs"):
            continue
        path = os.path.join(root, f)
        lines = open(path, encoding="utf-8", errors="replace").read().splitlines()
        tests = []
        for i, l in enumerate(lines):
            if l.strip() == "#[test]":
                j = i + 1
                attrs = []
                while j < len(lines) and lines[j].strip().startswith("#["):
                    attrs.append(lines[j].strip()); j += 1
                m = re.match(r"\s*fn\s+([A-Za-z0-9_]+)", lines[j])
                if m:
                    tests.append({"name": m.group(1), "line": j + 1, "should_panic": "#[should_panic]" in attrs})
        if tests:
            out[os.path.relpath(path, REF)] = tests
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json"), "w"), indent=1, sort_keys=True)
print({k: len(v) for k, v in out.items()}, sum(len(v) for v in out.values()))

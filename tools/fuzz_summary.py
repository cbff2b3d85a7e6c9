#!/usr/bin/env python3
"""Summary of tools/fuzz_end_to_end.py logs for profiles/rNN_fuzz_end_to_end.json:
python tools/fuzz_summary.py LOG [LOG ...]  ->  one JSON object per log on stdout."""
import sys, re, json
for path in sys.argv[1:]:
    e = dict(runs_compared=0, with_bidirectional_surface=0, with_dinput=0, with_ck_files=0, tokens_compared=0, failures=0,
             one_unit_off=0, ill_conditioned=0, reference_prints_nan=0, heating_rate_cancellation=0)
    for line in open(path):
        head = line.split("::")[0]
        if line.startswith("ok"):
            e["runs_compared"] += 1
            m = re.match(r"ok\s+(\d+) tokens, (\d+) off", line)
            e["one_unit_off"] += int(m.group(2))
            e["with_bidirectional_surface"] += bool(re.search(r"isalb=-?[789]\b", line))
            e["with_dinput"] += "&DINPUT" in line
            e["with_ck_files"] += "kdist=-1" in line
        elif line.startswith("FAIL"):
            e["failures"] += 1
        elif line.startswith("ill-conditioned"):
            e["ill_conditioned"] += 1
        elif line.startswith("skip") and "NaN" in head:
            e["reference_prints_nan"] += 1
        elif "cancellation" in head:
            e["heating_rate_cancellation"] += 1
        elif line.startswith("failures"):
            e["tokens_compared"] = int(line.split()[-1])
    print(json.dumps({path: e}))

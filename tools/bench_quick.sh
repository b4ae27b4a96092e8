#!/bin/bash
# short bench runs: prints the key numbers of the JSON line
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" 2>gpurun_out/bq.err | grep '^{"metric"' | tail -1 > gpurun_out/bq.json; python - "$@" <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/bq.json"))
except Exception as e:
    print("FAILED", sys.argv[1:], open("gpurun_out/bq.err").read()[-2000:]); sys.exit(0)
r=d["roofline"]
print(" ".join(sys.argv[1:]), "| value %.0f ingest %.0f steady %s | roofline frac %.4f ms/call %.3f scans/call %d block %s in-pipe %.3f ms/scan | parity %s" % (
  d["value"], d["value_with_ingest"]["value"], ("%.0f" % d["steady_state"]["value"]) if d.get("steady_state") else "-", r["frac"], r["kernel_ms"], r["scans_per_launch"], r["block_path"], r["in_pipeline"]["kernel_ms_per_scan"] or 0, d.get("parity")))
PY
}
for args in "$@"; do run $args; done

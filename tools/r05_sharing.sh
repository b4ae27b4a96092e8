#!/bin/bash
# profiles/r05_sharing_sensitivity.json: block path vs per-scan chain on windows chosen by the reference's rule
# (split_traintest.py:79-101), by speed bucket (live m/s, history lo-hi m/s), Lyft and nuScenes shape
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r05_sharing_sensitivity.jsonl
python tools/pp_block_probe.py --scans 16 --reps 3 --shards 2 --json-out gpurun_out/r05_sharing_sensitivity.jsonl > /dev/null 2>&1
for m in 3,3,5 3,5,10 3,10,15 5,3,5 5,5,10 8,3,15 8,5,10 8,10,15 12,5,10 12,10,15 15,3,15 15,10,15; do
  python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 --matched $m --json-out gpurun_out/r05_sharing_sensitivity.jsonl > /dev/null 2>&1
done
python tools/pp_block_probe.py --scans 16 --reps 3 --shards 2 --n 35000 --trav 20 --frames 16 --nusc --json-out gpurun_out/r05_sharing_sensitivity.jsonl > /dev/null 2>&1
for m in 5,3,8 8,3,15 12,8,15; do
  python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 --n 35000 --trav 20 --nusc --matched $m --json-out gpurun_out/r05_sharing_sensitivity.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r05_sharing_sensitivity.jsonl")]
for r in rows:
    sh = r["sharing"]
    print("%-5s %-10s union/members %.2f repeats %5.1f shared %.2f | block %6.1f us (%.3f) chain %6.1f us | parity %s" % (
        "nusc" if r["nusc"] else "lyft", r["matched"] or "i..i+F-1", sh["union_over_members"], sh.get("repeats_per_scan", 0.0),
        sh.get("shared_with_previous", 0.97), r["block_us_per_scan"], r["block_frac"], r["chain_us_per_scan"], r["parity_ok"]))
json.dump(dict(note="tools/r05_sharing.sh: 16 consecutive scans, block path (modest_pp_score_block) against the per-scan chain "
               "(modest_pp_score_frames_batch), windows chosen by the reference's rule (synth.make_shard_matched: live speed, "
               "history speed range in m/s at 5 Hz) -- first row: the frames i..i+F-1 windows every earlier number was measured on",
               rows=rows), open("gpurun_out/r05_sharing_sensitivity.json", "w"), indent=1)
PY

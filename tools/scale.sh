#!/bin/bash
# The scaling curve in one command, for the first machine that has more than one MI355X:
#     bash tools/scale.sh [steps] [warmup]
# runs `bench.py --gpus N` for N in 1 2 4 8 (bench.py starts the ranks itself through torch.distributed.run on 127.0.0.1,
# one process per GPU over RCCL), keeps the four JSON lines under gpurun_out/scale/, and checks what a scaling run must
# show: every line names N ranks on the "nccl" communicator, the strong-scaling entry timed its all-reduce, and the N = 1
# value of this sweep agrees with a plain `python bench.py` line within 5 %.  Efficiencies are NOT computed here beyond a
# printed table: the driver derives them from the per-N values.
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-20}
WARMUP=${2:-5}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
O=gpurun_out/scale
mkdir -p $O
timeout 900 python bench.py --steps $STEPS --warmup $WARMUP --no-secondary --no-cpu-baseline > $O/plain.json 2> $O/plain.err || { echo "plain bench.py failed"; tail -5 $O/plain.err; exit 1; }
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "N=$N: only $NGPU GPU(s) here -- skipped"; continue; fi
  timeout 1800 python bench.py --gpus $N --steps $STEPS --warmup $WARMUP --no-secondary --no-cpu-baseline > $O/n$N.json 2> $O/n$N.err \
    || { echo "bench.py --gpus $N failed"; tail -5 $O/n$N.err; exit 1; }
done
python - "$O" <<'PY'
import json, os, sys
O = sys.argv[1]
plain = json.load(open(os.path.join(O, "plain.json")))
rows, bad = [], []
for n in (1, 2, 4, 8):
    p = os.path.join(O, f"n{n}.json")
    if not os.path.exists(p):
        continue
    r = json.load(open(p))
    c = r["config"]
    if r["n_gpus"] != n or c.get("comm_world_size") != n:
        bad.append(f"N={n}: n_gpus {r['n_gpus']}, comm_world_size {c.get('comm_world_size')}")
    if n > 1 and c.get("comm_backend") != "nccl":
        bad.append(f"N={n}: comm_backend {c.get('comm_backend')!r}, expected 'nccl' (RCCL)")
    if n > 1 and not c.get("strong_allreduce_ms"):
        bad.append(f"N={n}: strong_allreduce_ms is {c.get('strong_allreduce_ms')!r}")
    rows.append((n, r["value"], r["ms_per_step"], c.get("strong_points_per_s"), c.get("strong_ms_per_step"), c.get("strong_allreduce_ms")))
if rows and rows[0][0] == 1 and abs(rows[0][1] / plain["value"] - 1.0) > 0.05:
    bad.append(f"N=1 of the sweep {rows[0][1]:.4g} points/s vs the plain line {plain['value']:.4g}: more than 5 % apart")
print(f"{'N':>2} {'weak points/s':>14} {'ms/step':>8} {'weak eff':>8} | {'strong points/s':>15} {'ms/step':>8} {'all-reduce ms':>13} {'speed-up':>8}")
for n, v, ms, sv, sms, ar in rows:
    print(f"{n:>2} {v:14.4g} {ms:8.4f} {v / (n * rows[0][1]):8.3f} | {sv or 0:15.4g} {sms or 0:8.4f} {ar or 0:13.4f} {(sv or 0) / (rows[0][3] or 1):8.2f}")
for b in bad:
    print("CHECK FAILED:", b)
sys.exit(1 if bad else 0)
PY

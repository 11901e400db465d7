mkdir -p gpurun_out/pf
for N in 2 8; do timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2961$N bench.py --gpus $N --steps 3 --warmup 1 --no-extras > gpurun_out/pf/n$N.json 2> gpurun_out/pf/n$N.err; echo "N=$N rc=$?"; python3 - gpurun_out/pf/n$N.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pr=d["config"]["corner_turn_probes"]
print(d["value"], pr["chosen"][:40]); print([ (p["form"][:34], p.get("passed"), p.get("this_rank_status"), p.get("seconds"), p.get("available"), p.get("correct")) for p in pr["probes"]])
PY
done
STARKCORE_TEST_PREFLIGHT_DIES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --steps 3 --warmup 1 --no-extras > gpurun_out/pf/dies.json 2> gpurun_out/pf/dies.err; echo "dies rc=$?"; grep "pre-flight" gpurun_out/pf/dies.err | head -3; python3 - gpurun_out/pf/dies.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pr=d["config"]["corner_turn_probes"]
print(d["value"], pr["chosen"][:60]); print([ (p["form"][:34], p.get("passed"), p.get("this_rank_status"), p.get("available")) for p in pr["probes"]])
PY

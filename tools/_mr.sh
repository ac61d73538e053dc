cd tests
python - <<'PY'
import json, os, random, subprocess, sys
REPO = os.path.dirname(os.getcwd())
ARGS = ["--res", "256", "--steps", "10", "--warmup", "4", "--sweeps", "0", "--no-cpu"]
def single(extra=()):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + ARGS + list(extra), capture_output=True, text=True)
    return json.loads(out.stdout.strip().splitlines()[-1])
def multi(world, extra=(), env=None):
    port = random.randint(20000, 50000)
    procs = []
    for r in range(world):
        e = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", FS_FAKE_PORT=str(port), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 100), **(env or {}))
        procs.append(subprocess.Popen([sys.executable, "bench_socket_worker.py", "--gpus", str(world)] + ARGS + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        if p.returncode: print("ERR", se[-1500:])
    return json.loads(outs[0][0].strip().splitlines()[-1])
s = single(); print("single graph", s["state_checksum"], s["config"]["launch"])
s2 = single(["--no-graph"]); print("single eager", s2["state_checksum"])
for w in (2,):
    for extra in ((), ("--no-tape",)):
        d = multi(w, extra); print(w, extra, d["state_checksum"], d["config"]["launch"])
    d = multi(w, (), {"FS_NO_HOIST": "1"}); print(w, "nohoist", d["state_checksum"], d["config"]["launch"])
PY

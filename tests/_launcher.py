"""Child-process launcher for the GPU box: started by tests/conftest.py BEFORE anything in the pytest process touches
the GPU, it never touches the GPU itself, so its fork+exec of worker processes is always allowed (a process that
has initialised HIP must not exec; the pytest process has, by the time the multi-rank test runs).
Protocol: one JSON request per line on stdin {"argv": [...], "env": {...}, "n": ranks, "timeout": s, "raw": bool} ->
one JSON reply per line {"rc": [..], "out": [tail per rank]}."""
import json
import os
import subprocess
import sys


def main():
    for line in sys.stdin:
        req = json.loads(line)
        procs = []
        for r in range(req["n"]):
            env = dict(os.environ)
            env.update(req["env"])
            if not req.get("raw"):  # "raw": a plain child (e.g. bench.py --gpus N, which starts its own ranks)
                env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(req["n"]))
            procs.append(subprocess.Popen(req["argv"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        rc, out = [], []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=req.get("timeout", 600))
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
                o += "\n[launcher] timeout"
            rc.append(p.returncode)
            out.append(o[-60000:])
        sys.stdout.write(json.dumps({"rc": rc, "out": out}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()

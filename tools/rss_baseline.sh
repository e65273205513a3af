python - <<'PY'
import os, resource, subprocess, sys
sys.path.insert(0, ".")
import numpy as np
import crass_amd as ca
ca.load()
open("/tmp/tiny.fa","wb").write(b">a\nACGT\n")
def run(args, env=None):
    e = dict(os.environ); e.update(env or {})
    os.makedirs("/tmp/o", exist_ok=True)
    p = subprocess.Popen(["crass_amd/crass-hip", "-g", "-o", "/tmp/o"] + args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=e)
    _, st, ru = os.wait4(p.pid, 0)
    return ru.ru_maxrss / 1024.0
print("tiny input: peak RSS %.0f MB" % run(["/tmp/tiny.fa"]))
PY

#!/usr/bin/env python3
"""Static audit of this directory against the CURRENT library (VERDICT r3 task 8): every script must byte-compile, and every WCT_*
environment variable / wct_debug_set key it uses must still exist in csrc/, wct_hip/ or bench.py.  Prints one row per script; exit 1 on a
stale reference.  (Run by tests/test_abi_cpu.py::test_experiment_scripts_reference_live_switches.)"""
import glob, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
src = " ".join(open(f, errors="ignore").read() for f in glob.glob(os.path.join(REPO, "collaborative-distillation_amd", "csrc", "*")) +
               glob.glob(os.path.join(REPO, "collaborative-distillation_amd", "wct_hip", "*.py")) + [os.path.join(REPO, "bench.py")])
envs = set(re.findall(r"(WCT_[A-Z0-9_]+)", src))
keys = set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', src))
bad = 0
for f in sorted(glob.glob(os.path.join(HERE, "*"))):
    b = os.path.basename(f)
    if not b.endswith((".py", ".sh", ".hip")) or b == "audit.py":
        continue
    t = open(f, errors="ignore").read()
    used = set(re.findall(r"(WCT_[A-Z0-9_]+)", t)) - {"WCT_DEBUG"}
    stale_env = sorted(u for u in used if u not in envs)
    dk = set(re.findall(r'debug_set\("([a-z0-9_]+)"', t)) | set(re.findall(r"--debug-set ([a-z0-9_]+)=", t))
    stale_key = sorted(k for k in dk if k not in keys)
    ok = subprocess.call([sys.executable, "-m", "py_compile", f], stderr=subprocess.DEVNULL) == 0 if b.endswith(".py") else True
    status = "ok" if ok and not stale_env and not stale_key else "STALE"
    bad += status != "ok"
    print("%-28s %-6s %s" % (b, status, " ".join(stale_env + stale_key + ([] if ok else ["does-not-compile"]))))
sys.exit(1 if bad else 0)

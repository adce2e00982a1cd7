"""ISA check: every s_barrier that is reached - walking BACKWARDS through straight-line code and fall-through labels - from an LDS write (ds_write* / ds_store*)
without an `s_waitcnt ... lgkmcnt(0)` in between.  hipcc (ROCm 7.2, gfx950) drops the LDS wait of __syncthreads()'s release fence (it assumes LDS operations of all
waves are totally ordered); on MI355X a ds_write issued right before the barrier can still be in flight when another SIMD's wave reads the location behind it
(round 6: one staged tile in ~1000 launches of the f32e ConvGRU step read stale data).  Usage: python tools/scan_barrier_waits.py [file.hip ...] (default: all of csrc)."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", path, "-o", out], capture_output=True, text=True)
        if r.returncode:
            return None
        lines = open(out).read().split("\n")
    cur, bad, total = None, {}, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        if l.strip() != "s_barrier" or cur is None:
            continue
        total += 1
        j = i - 1
        while j >= 0:
            t = lines[j].strip()
            j -= 1
            if not t or t.startswith(";") or t.startswith(".") and not t.endswith(":"):
                continue
            if re.match(r"^_Z\w+:", t) or t.startswith("s_endpgm") or t.startswith("s_branch") or t.startswith("s_setpc"):
                break                       # start of the function / an unconditional jump above: this path ends
            if t.startswith("s_waitcnt") and ("lgkmcnt(0)" in t or re.match(r"s_waitcnt\s+0x0\b|s_waitcnt\s+0\b", t)):
                break
            if t.startswith("s_barrier"):
                break
            if re.match(r"ds_(write|store)", t):
                bad[cur] = bad.get(cur, 0) + 1
                break
    return total, bad


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "satflow_amd", "csrc", "*.hip")))
    worst = 0
    for f in files:
        r = scan(f)
        if r is None:
            print(os.path.basename(f), "did not compile"); continue
        total, bad = r
        n = sum(bad.values()); worst += n
        print(f"{os.path.basename(f):36s} barriers {total:5d}   reached from an LDS write without lgkmcnt(0): {n:4d} in {len(bad)} kernels")
        for k, v in list(bad.items())[:3]:
            print("      ", k[:120], v)
    sys.exit(1 if worst else 0)

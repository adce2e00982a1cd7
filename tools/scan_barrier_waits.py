"""ISA check: every s_barrier that is reached - along any path of the kernel's control-flow graph, back edges included - from an LDS write (ds_write* / ds_store*)
without an `s_waitcnt ... lgkmcnt(0)` in between.  hipcc (ROCm 7.2, gfx950) drops the LDS wait of __syncthreads()'s release fence (it assumes LDS operations of all
waves are totally ordered); on MI355X a ds_write issued right before the barrier can still be in flight when another SIMD's wave reads the location behind it
(round 6: one staged tile in ~1000 launches of the f32e ConvGRU step read stale data).  Usage: python tools/scan_barrier_waits.py [file.hip ...] (default: all of csrc)."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernels(lines):
    """Split the assembly into functions: name -> list of instruction / label lines."""
    cur, out = None, {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); out[cur] = []
            continue
        t = l.strip()
        if cur is None or not t or t.startswith(";"):
            continue
        if t.startswith(".") and not t.endswith(":"):
            if t.startswith(".section") or t.startswith(".rodata") or t.startswith(".amdhsa_kernel"):
                cur = None
            continue
        out[cur].append(t.split(";")[0].strip())
    return out


def _scan_function(ins):
    """Forward dataflow over the function's control-flow graph: 'an LDS write of this wave may still be in flight' is set by ds_write* / ds_store*, cleared by an
    s_waitcnt with lgkmcnt(0) (or the all-zero form); a barrier reached in that state is a hit.  Back edges included (iterated to a fixed point)."""
    blocks, cur, label_of = [], [], {}
    for t in ins:
        if t.endswith(":"):
            if cur:
                blocks.append(cur)
            cur = [t]
            continue
        cur.append(t)
        if t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append(cur); cur = []
    if cur:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b and b[0].endswith(":"):
            label_of[b[0][:-1]] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1] if b else ""
        s_ = []
        if last.startswith("s_branch"):
            tgt = last.split()[-1]
            if tgt in label_of: s_.append(label_of[tgt])
        elif last.startswith("s_cbranch"):
            tgt = last.split()[-1]
            if tgt in label_of: s_.append(label_of[tgt])
            if i + 1 < len(blocks): s_.append(i + 1)
        elif last.startswith(("s_endpgm", "s_setpc")):
            pass
        elif i + 1 < len(blocks):
            s_.append(i + 1)
        succ.append(s_)
    state_in = [False] * len(blocks)
    hits, barriers = 0, 0
    changed, first = True, True
    while changed:
        changed, hits, barriers = False, 0, 0
        for i, b in enumerate(blocks):
            pend = state_in[i]
            for t in b:
                if re.match(r"ds_(write|store)", t):
                    pend = True
                elif t.startswith("s_waitcnt") and ("lgkmcnt(0)" in t or re.match(r"s_waitcnt\s+(0x0|0)\b", t)):
                    pend = False
                elif t == "s_barrier":
                    barriers += 1
                    if pend:
                        hits += 1
            for j in succ[i]:
                if pend and not state_in[j]:
                    state_in[j] = True; changed = True
    return barriers, hits


def scan(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", path, "-o", out], capture_output=True, text=True)
        if r.returncode:
            return None
        lines = open(out).read().split("\n")
    total, bad = 0, {}
    for name, ins in _kernels(lines).items():
        b, h = _scan_function(ins)
        total += b
        if h:
            bad[name] = h
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

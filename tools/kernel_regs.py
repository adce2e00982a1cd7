"""Register / LDS / scratch use of every kernel of a HIP source (hipcc -S for gfx950, metadata of the code object): python tools/kernel_regs.py <file.hip> [substring]
Runs HERE (hipcc cross-compiles).  Used to check which kernels can share a CU (512 VGPRs per SIMD lane, 8-register granules, 160 KB LDS)."""
import os, re, subprocess, sys, tempfile

src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *extra, "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
for b in s.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if sub and sub not in dem:
        continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", b).group(1)
    print(f"vgpr={g('vgpr_count'):>4} agpr={b.splitlines()[0].strip():>3} sgpr={g('sgpr_count'):>3} lds={g('group_segment_fixed_size'):>6} scratch={g('private_segment_fixed_size'):>4}  {dem[:150]}")

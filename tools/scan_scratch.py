"""Which compiled kernels touch scratch memory (private segment) or spill?  Reads the gfx950 code objects out of the
built objects under gcm/_lib (llvm-objcopy + clang-offload-bundler + llvm-readelf --notes).  Found in round 4: an array
of HIP's float4 STRUCT with 16 entries is not promoted to registers by hipcc 7.2 (272 bytes of scratch per lane); the
compiler's own ext_vector_type(4) is."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graph-conv-memory_amd", "gcm", "_lib")


def main():
    tmp = tempfile.mkdtemp()
    for f in sorted(glob.glob(os.path.join(LIB, "*.o"))):
        b = os.path.basename(f)[:-2]
        fat, co = os.path.join(tmp, b + ".fat"), os.path.join(tmp, b + ".co")
        if subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", f],
                          capture_output=True).returncode:
            continue
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if not os.path.exists(co):
            continue
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count", txt)[1:]:
            n = re.search(r"\.name:\s+(\S+)", blk)
            p = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            if n and p and (int(p.group(1)) > 0 or "-a" in sys.argv):
                name = subprocess.run(["c++filt", n.group(1)], capture_output=True, text=True).stdout.strip()
                print(f"{b}: {name[:110]} | scratch {p.group(1)} B, {v.group(1)} vgpr, {sp.group(1) if sp else '?'} spilled")


if __name__ == "__main__":
    main()

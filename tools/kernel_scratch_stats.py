import subprocess, os, time, re, sys
sys_path_fix = __import__("sys").path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cudadepthmapintegration_amd import build as b
out="/tmp/dis/tile_fast3.s"
cmd=[b.hipcc_path()]+b.COMMON_FLAGS+b.HIP_FLAGS+["-DDMI_FAST_BUILD"]+sys.argv[1:]+["--cuda-device-only","-S",os.path.join(b.CSRC, os.environ.get("TILE_SRC","fusion_tile.hip")),"-o",out]
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
t=open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
    n=m.group(1); body=m.group(2)
    sc=re.search(r"private_segment_fixed_size (\d+)",body).group(1)
    k=re.search(r"fuse_tile_kernelI(\w\wLi\d+E.*?)EvNS",n)
    if k: print(k.group(1).replace("ELi1ELi1E","").replace("ELb",""), "scratch",sc)

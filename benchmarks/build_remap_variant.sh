#!/bin/bash
# An experiment build of the library with other -D knobs for the fused-mean part of csrc/remap.hip (float64, four fields only;
# the sweep part is the tree's remap.o):
#   benchmarks/build_remap_variant.sh NAME [-DMEAN_KOUT=8 ...]   ->  gpurun_variants/libfv3hip_NAME.so
# Select it at run time with FV3HIP_LIBRARY=... (benchmarks/block_mean_timing.py, remap_sweep_timing.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/gpurun_variants
cd $R/fv3net_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -Wno-inline-asm -Wno-unused-function \
    -mllvm -pragma-unroll-threshold=262144 -DFV3HIP_REMAP_SUBSET -DFV3HIP_REMAP_PART_MEAN "$@" -Rpass-analysis=kernel-resource-usage -c remap.hip -o /tmp/remap_$name.o 2> /tmp/remap_$name.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_variants/libfv3hip_$name.so capi.o coarsen.o vertical.o remap.o /tmp/remap_$name.o mlp.o mlp_bf16x3.o emulation.o local.o fit.o
python3 - $name <<'PY'
import re,subprocess,sys
t=open(f'/tmp/remap_{sys.argv[1]}.log').read()
blocks=re.split(r'remark: [^\n]*Function Name: ',t)[1:]
names=[b.split('\n')[0].strip().split()[0] for b in blocks]
dn=subprocess.run(['c++filt']+names,capture_output=True,text=True).stdout.strip().split('\n')
for b,d in zip(blocks,dn):
    g=lambda k: re.search(k+r': (\d+)',b).group(1)
    if 'sweep_kernel' in d:
        print(d.replace('void fv3hip::(anonymous namespace)::','')[:62],'VGPR',g('VGPRs'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'))
PY

# generates mfma_ubench.hip: slots of 8 independent fp32 32x32x2 MFMAs with side work between them
modes = {
 0: [],                                   # MFMAs only
 1: ["valu"]*8,                           # 8 VALU
 2: ["valu"]*32,                          # 32 VALU
 3: ["dsr128","dsr128","dsr32","wait0"],  # A/b reads of a slot (wait at end)
 4: ["dsw128"],                           # one LDS commit
 5: ["dsr128","dsr128","dsr32"]+["valu"]*8+["dsw128","wait1"],  # finish slot mimic
 6: ["salu"]*16,
 7: ["valu_dep"]*8,                       # dependent chain VALU
 8: ["nop"]*32,
 9: ["vmov"]*32,
 11: ["G", "valu"]*0 + ["valu@0"]*32,     # 32 VALU in one group after M0
 12: ["pk"]*16,
 13: ["pk@0"]*16,
 14: ["gld"]*4 + ["waitvm"],
 15: ["accw"]*8,
 16: ["mad64"]*4,
 17: ["valu@%d" % i for i in range(8)],
 18: ["valu@0"]*8,
 19: ["valu@0"]*16,
 20: ["valu@0"]*4,
 21: ["valu@0"]*1,
 22: ["valu@0", "valu@4"],
 23: ["pkfma@0"]*8,
 24: ["valu@7"]*8,
 25: ["bar@0"],
 26: ["bar@7"],
 27: ["dsr128@0","dsr128@0","bar@0","dsr128@0","dsr128@0"],
 28: ["gld@0","gld@0","gld@0","gld@0"],
 29: ["gld@0","gld@0","gld@0","gld@0","waitvm@7","dsw128@7","dsw128@7","dsw128@7","dsw128@7"],
 30: ["vaddco@0"]*4,
 31: ["dsr128@0","dsr128@0","waitl@3"],
 32: ["dsr128@0","dsr128@0","waitl@1"],
 40: [],
 41: [],
 42: ["dsr128@0", "dsw128@2", "valu@4", "valu@4"],
}
def body(side, chain=False, chain2=False):
    # distribute side ops after MFMAs round-robin (up to 5 after each)
    out = []
    per = [[] for _ in range(8)]
    for i, op in enumerate(side):
        if "@" in op:
            o, t = op.split("@"); per[int(t)].append(o)
        elif op.startswith("wait"):
            per[7].append(op)
        else:
            per[min(i // 5, 7)].append(op)
    vi = 0
    for t in range(8):
        tt = 0 if chain else (t % 2 if chain2 else t)
        out.append(f"v_mfma_f32_32x32x2_f32 a[{16*tt}:{16*tt+15}], v{2+t}, v1, a[{16*tt}:{16*tt+15}]")
        for op in per[t]:
            if op == "valu":
                out.append(f"v_sub_f32 v{20+vi%16}, v{20+vi%16}, v40"); vi += 1
            elif op == "pk":
                out.append(f"v_pk_mul_f32 v[{20+2*(vi%8)}:{21+2*(vi%8)}], v[{20+2*(vi%8)}:{21+2*(vi%8)}], v[40:41]"); vi += 1
            elif op == "pkfma":
                out.append(f"v_pk_fma_f32 v[{20+2*(vi%8)}:{21+2*(vi%8)}], v[{20+2*(vi%8)}:{21+2*(vi%8)}], v[40:41], v[40:41]"); vi += 1
            elif op == "gld":
                out.append(f"global_load_dwordx4 v[{20+4*(vi%4)}:{23+4*(vi%4)}], v[54:55], off offset:{16*(vi%4)}"); vi += 1
            elif op == "waitvm":
                out.append("s_waitcnt vmcnt(0)")
            elif op == "accw":
                out.append(f"v_accvgpr_write_b32 a{vi%16}, v40"); vi += 1
            elif op == "mad64":
                out.append(f"v_mad_u64_u32 v[{20+2*(vi%8)}:{21+2*(vi%8)}], s[22:23], v40, v41, v[56:57]"); vi += 1
            elif op == "bar":
                out.append("s_waitcnt lgkmcnt(0)"); out.append("s_barrier")
            elif op == "waitl":
                out.append("s_waitcnt lgkmcnt(0)")
            elif op == "vaddco":
                out.append("v_add_co_u32 v20, vcc, s22, v40"); out.append("v_addc_co_u32 v21, vcc, 0, v41, vcc")
            elif op == "valu_dep":
                out.append("v_sub_f32 v20, v20, v40")
            elif op == "vmov":
                out.append(f"v_mov_b32 v{20+vi%16}, v40"); vi += 1
            elif op == "salu":
                out.append("s_add_u32 s20, s20, 1")
            elif op == "nop":
                out.append("s_nop 0")
            elif op == "dsr128":
                out.append(f"ds_read_b128 v[{44+4*(vi%2)}:{47+4*(vi%2)}], v41 offset:{1024*(vi%2)}"); vi += 1
            elif op == "dsr32":
                out.append("ds_read_b32 v52, v41 offset:4096")
            elif op == "dsw128":
                out.append("ds_write_b128 v42, v[56:59]")
            elif op in ("wait0", "wait0_first"):
                out.append("s_waitcnt lgkmcnt(0)")
            elif op == "wait1":
                out.append("s_waitcnt lgkmcnt(1)")
    return out
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '']
for m, side in modes.items():
    b = body(side, chain=(m == 40 or m == 42), chain2=(m == 41))
    asm = "\\n\\t".join(b)
    clob = ",".join([f'"a{i}"' for i in range(128)] + [f'"v{i}"' for i in range(1, 60)] + ['"s20"', '"memory"'])
    src.append(f'''__global__ __launch_bounds__(256, 1) void k{m}(long long *out, int iters, const float *buf) {{
    __shared__ float lds[16384];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    asm volatile("v_mov_b32 v1, 1.0\\n\\tv_mov_b32 v40, 0.5\\n\\tv_lshlrev_b32 v41, 4, %0\\n\\tv_add_u32 v42, 0x8000, v41\\n\\tv_mov_b32 v54, %1\\n\\tv_mov_b32 v55, %2" :: "v"(threadIdx.x), "v"((unsigned)((unsigned long long)(buf + threadIdx.x * 16) & 0xffffffffu)), "v"((unsigned)((unsigned long long)(buf + threadIdx.x * 16) >> 32)) : "v1","v40","v41","v42","v54","v55");
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {{
        asm volatile("{asm}" ::: {clob});
    }}
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}}''')
src.append('''int main() {
    long long *d; hipMalloc(&d, 256 * 8); long long h[256]; const int iters = 4000; float *buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);''')
for m in modes:
    src.append(f'''    k{m}<<<256, 256>>>(d, iters, buf); hipDeviceSynchronize(); k{m}<<<256, 256>>>(d, iters, buf); hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
    {{ double s = 0; for (int i = 0; i < 256; ++i) s += h[i]; printf("mode {m}: %.1f cycles/slot (ideal 512)  side ops: {len(modes[m])}\\n", s / 256 / iters); }}''')
src.append("    return 0;\n}")
open("mfma_ubench.hip", "w").write("\n".join(src))

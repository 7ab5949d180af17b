// issue_rates.hip -- what does ONE wavefront alone on a SIMD pay per instruction on gfx950?  (diagnostic, not product)
// Each test runs ITER iterations of a block of 16 identical-type instructions between two s_memtime reads
// (shader-clock ticks) and two wall_clock64() reads (100 MHz), so the output also gives the shader clock.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_rates.hip -o build/issue_rates && build/issue_rates [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 4096
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

struct Res { unsigned long long cyc, wall; };

#define TEST_KERNEL(name, body)                                                                     \
    __global__ __launch_bounds__(64) void name(Res *out, float *sink, const int *idx)               \
    {                                                                                               \
        __shared__ int lds[256];                                                                    \
        lds[threadIdx.x] = (threadIdx.x * 4 + 4) & 255;                                             \
        lds[threadIdx.x + 64] = 0; lds[threadIdx.x + 128] = 0; lds[threadIdx.x + 192] = 0;          \
        float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;                                        \
        int s0 = idx[0], s1 = idx[1], s2 = 3, s3 = 4;                                               \
        int v = threadIdx.x * 4;                                                                    \
        const int *gp = idx + idx[2];                                                               \
        __syncthreads();                                                                            \
        unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();                  \
        for (int it = 0; it < ITER; it++) { body }                                                  \
        unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();                  \
        if (threadIdx.x == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].wall = w1 - w0; }    \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + s0 + s1 + s2 + s3 + v + (float)(size_t)gp; \
    }

TEST_KERNEL(k_valu_dep, asm volatile(R16("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));)
TEST_KERNEL(k_valu_ind, asm volatile(R4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f));)
TEST_KERNEL(k_salu_dep, asm volatile(R16("s_add_u32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc");)
TEST_KERNEL(k_salu_ind, asm volatile(R4("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");)
TEST_KERNEL(k_mix_sv, asm volatile(R4("s_add_u32 %0, %0, 1\n v_add_f32 %2, %2, %3\n s_add_u32 %1, %1, 1\n v_add_f32 %3, %3, %2\n") : "+s"(s0), "+s"(s1), "+v"(a), "+v"(b) : : "scc");)
TEST_KERNEL(k_snop, asm volatile(R16("s_nop 0\n"));)
TEST_KERNEL(k_dpp_dep, asm volatile(R4("v_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 0\n v_add_f32 %3, %3, %2\n") : "+v"(a), "+v"(b) : "v"(c), "v"(d));)
TEST_KERNEL(k_readlane_use, asm volatile(R4("v_readlane_b32 %0, %2, 3\n s_nop 1\n v_mul_f32 %2, %0, %2\n s_nop 0\n") : "+s"(s0) : "s"(s1), "v"(a) );)
TEST_KERNEL(k_cmp_sand, asm volatile(R4("v_cmp_lt_f32 vcc, %1, %2\n s_and_b64 %0, vcc, exec\n v_cndmask_b32 %1, 0, %1, %0\n s_nop 0\n") : "+s"(*(unsigned long long *)&s0) : "v"(a), "v"(b) : "vcc", "scc");)
TEST_KERNEL(k_lds_chain, asm volatile(R16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(v));)
TEST_KERNEL(k_lds_issue, asm volatile(R16("ds_read_b32 %1, %0\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(v), "v"(a));)
TEST_KERNEL(k_vmem_issue, asm volatile(R4("global_load_dword %1, %0, %2\n") "s_waitcnt vmcnt(0)\n" : : "v"(v), "v"(a), "s"(gp));)
TEST_KERNEL(k_vmem_issue4, asm volatile(R4("global_load_dwordx4 v[100:103], %0, %1\n") "s_waitcnt vmcnt(0)\n" : : "v"(v), "s"(gp) : "v100", "v101", "v102", "v103");)
TEST_KERNEL(k_vmem_chain, asm volatile(R4("global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n") : "+v"(v) : "s"(gp));)
TEST_KERNEL(k_branch_nt, asm volatile(R16("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n") "1:\n" : : "s"(s0) : "scc");)
TEST_KERNEL(k_branch_tk, asm volatile(R4("s_branch 1f\n s_nop 0\n 1:\n s_branch 2f\n s_nop 0\n 2:\n s_branch 3f\n s_nop 0\n 3:\n s_branch 4f\n s_nop 0\n 4:\n"));)
TEST_KERNEL(k_writelane, asm volatile(R16("v_writelane_b32 %0, %1, m0\n") : "+v"(a) : "s"(s0) : "m0");)
TEST_KERNEL(k_smul, asm volatile(R16("s_mul_i32 %0, %0, %1\n") : "+s"(s0) : "s"(s1));)
TEST_KERNEL(k_ff1, asm volatile(R16("s_ff1_i32_b64 %0, %1\n") : "=s"(s0) : "s"(*(unsigned long long *)&s2));)


// ---- second batch: how far apart must producer and consumer be, and what do loads cost with fewer lanes?
TEST_KERNEL(k_br_d0, asm volatile(R4("s_cmp_eq_u32 %0, 12345\n s_cbranch_scc1 1f\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n") "1:\n" : : "s"(s0), "v"(a), "v"(b) : "scc");)
TEST_KERNEL(k_br_d2, asm volatile(R4("s_cmp_eq_u32 %0, 12345\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n s_cbranch_scc1 1f\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n") "1:\n" : : "s"(s0), "v"(a), "v"(b) : "scc");)
TEST_KERNEL(k_br_d4, asm volatile(R4("s_cmp_eq_u32 %0, 12345\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n s_cbranch_scc1 1f\n") "1:\n" : : "s"(s0), "v"(a), "v"(b) : "scc");)
TEST_KERNEL(k_br_vccz, asm volatile(R4("v_cmp_lt_f32 vcc, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n s_cbranch_vccz 1f\n") "1:\n" : : "s"(s0), "v"(a), "v"(b) : "vcc");)
TEST_KERNEL(k_cmp_and_d0, asm volatile(R4("v_cmp_lt_f32 vcc, %1, %2\n s_and_b64 %0, vcc, exec\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n") : "+s"(*(unsigned long long *)&s0) : "v"(a), "v"(b) : "vcc", "scc");)
TEST_KERNEL(k_cmp_and_d2, asm volatile(R4("v_cmp_lt_f32 vcc, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n s_and_b64 %0, vcc, exec\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n") : "+s"(*(unsigned long long *)&s0) : "v"(a), "v"(b) : "vcc", "scc");)
TEST_KERNEL(k_cmp_and_d4, asm volatile(R4("v_cmp_lt_f32 vcc, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %1, %1, %2\n s_and_b64 %0, vcc, exec\n") : "+s"(*(unsigned long long *)&s0) : "v"(a), "v"(b) : "vcc", "scc");)
TEST_KERNEL(k_sand_cnd_d0, asm volatile(R4("s_and_b64 %0, %0, exec\n v_cndmask_b32 %1, 0, %1, %0\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n") : "+s"(*(unsigned long long *)&s0) : "v"(a), "v"(b) : "scc");)
TEST_KERNEL(k_rdl_salu_d0, asm volatile(R4("v_readlane_b32 %0, %1, 3\n s_add_u32 %0, %0, 1\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n") : "+s"(s0) : "v"(a), "v"(b) : "scc");)
TEST_KERNEL(k_rdl_salu_d4, asm volatile(R4("v_readlane_b32 %0, %1, 3\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n s_add_u32 %0, %0, 1\n") : "+s"(s0) : "v"(a), "v"(b) : "scc");)
TEST_KERNEL(k_salu_valu_d0, asm volatile(R4("s_add_u32 %0, %0, 1\n v_add_u32 %1, %0, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n") : "+s"(s0) : "v"(v), "v"(b) : "scc");)
TEST_KERNEL(k_vmem_l16, asm volatile("s_mov_b64 exec, 0xffff\n" R4("global_load_dword %1, %0, %2\n") "s_waitcnt vmcnt(0)\n s_mov_b64 exec, -1\n" : : "v"(v), "v"(a), "s"(gp));)
TEST_KERNEL(k_vmem_l32, asm volatile("s_mov_b64 exec, 0xffffffff\n" R4("global_load_dword %1, %0, %2\n") "s_waitcnt vmcnt(0)\n s_mov_b64 exec, -1\n" : : "v"(v), "v"(a), "s"(gp));)
TEST_KERNEL(k_vmem_l48, asm volatile("s_bfm_b64 exec, 48, 0\n" R4("global_load_dword %1, %0, %2\n") "s_waitcnt vmcnt(0)\n s_mov_b64 exec, -1\n" : : "v"(v), "v"(a), "s"(gp));)
TEST_KERNEL(k_vmem_nowait, asm volatile(R4("global_load_dword %1, %0, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %3, %3, %3\n v_add_f32 %3, %3, %3\n") "s_waitcnt vmcnt(0)\n" : : "v"(v), "v"(a), "s"(gp), "v"(b));)
TEST_KERNEL(k_vmem_x2, asm volatile(R4("global_load_dwordx2 v[100:101], %0, %1\n") "s_waitcnt vmcnt(0)\n" : : "v"(v), "s"(gp) : "v100", "v101");)
TEST_KERNEL(k_smem, asm volatile(R4("s_load_dwordx4 s[80:83], %0, 0x0\n") "s_waitcnt lgkmcnt(0)\n" : : "s"(gp) : "s80", "s81", "s82", "s83");)
TEST_KERNEL(k_smem_chain, asm volatile(R4("s_load_dword s80, %0, 0x0\n s_waitcnt lgkmcnt(0)\n") : : "s"(gp) : "s80");)
TEST_KERNEL(k_ds_read2, asm volatile(R16("ds_read2_b32 v[100:101], %0 offset1:1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(v) : "v100", "v101");)
TEST_KERNEL(k_ds_write, asm volatile(R16("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(v), "v"(a));)
TEST_KERNEL(k_loop_tk, asm volatile("s_mov_b32 s80, 16\n 1:\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n s_sub_u32 s80, s80, 1\n s_cmp_lg_u32 s80, 0\n s_cbranch_scc1 1b\n" : "+v"(a) : "v"(b) : "s80", "scc");)
TEST_KERNEL(k_setprio, asm volatile("s_setprio 3\n" R16("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));)


// ---- third batch: what does ISSUING a vector load cost (16 in a row, then one wait: latency is shared by 16)?
#define LD16(instr, regs) R4(instr " " LDREG(100, regs) ", %0, %1\n" instr " " LDREG(104, regs) ", %0, %1\n" instr " " LDREG(108, regs) ", %0, %1\n" instr " " LDREG(112, regs) ", %0, %1\n")
#define LDREG(a, regs) LDREG_##regs(a)
#define LDREG_1(a) "v" #a
#define LDREG_4(a) "v[" #a ":" #a "+3]"
TEST_KERNEL(k_vm16_same, asm volatile(LD16("global_load_dword", 1) "s_waitcnt vmcnt(0)\n" : : "v"(0), "s"(gp) : "v100", "v104", "v108", "v112");)
TEST_KERNEL(k_vm16_coal, asm volatile(LD16("global_load_dword", 1) "s_waitcnt vmcnt(0)\n" : : "v"(v), "s"(gp) : "v100", "v104", "v108", "v112");)
TEST_KERNEL(k_vm16_scat, asm volatile(LD16("global_load_dword", 1) "s_waitcnt vmcnt(0)\n" : : "v"(v * 64), "s"(gp) : "v100", "v104", "v108", "v112");)
TEST_KERNEL(k_vm16_scat8, asm volatile(LD16("global_load_dword", 1) "s_waitcnt vmcnt(0)\n" : : "v"((v >> 5) * 256 + (v & 31)), "s"(gp) : "v100", "v104", "v108", "v112");)
TEST_KERNEL(k_vm16_x4_scat, asm volatile(LD16("global_load_dwordx4", 4) "s_waitcnt vmcnt(0)\n" : : "v"(v * 64 + 4), "s"(gp) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");)
TEST_KERNEL(k_vm16_x4_coal, asm volatile(LD16("global_load_dwordx4", 4) "s_waitcnt vmcnt(0)\n" : : "v"(v * 4), "s"(gp) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");)
TEST_KERNEL(k_vm_mix, asm volatile(R4("global_load_dword v100, %0, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n v_add_f32 %2, %2, %2\n") "s_waitcnt vmcnt(0)\n" : : "v"(v * 64), "s"(gp), "v"(b) : "v100");)
TEST_KERNEL(k_sm16, asm volatile(R4("s_load_dwordx8 s[80:87], %0, 0x0\n s_load_dwordx8 s[88:95], %0, 0x40\n s_load_dwordx8 s[80:87], %0, 0x80\n s_load_dwordx8 s[88:95], %0, 0xc0\n") "s_waitcnt lgkmcnt(0)\n" : : "s"(gp) : "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");)

typedef void (*KF)(Res *, float *, const int *);
struct T { const char *name; KF f; int per_iter; };

int main(int argc, char **argv)
{
    int blocks = argc > 1 ? atoi(argv[1]) : 1;
    Res *d_out; float *d_sink; int *d_idx;
    hipMalloc(&d_out, sizeof(Res) * blocks); hipMalloc(&d_sink, sizeof(float) * 64 * blocks); hipMalloc(&d_idx, 1 << 20); hipMemset(d_idx, 0, 1 << 20);
    std::vector<int> h(1024, 0); h[0] = 1; h[1] = 2; h[2] = 16;
    hipMemcpy(d_idx, h.data(), 4096, hipMemcpyHostToDevice);
    T tests[] = {{"v_add_f32 dependent", k_valu_dep, 16}, {"v_add_f32 4 independent chains", k_valu_ind, 16}, {"s_add_u32 dependent", k_salu_dep, 16},
                 {"s_add_u32 independent", k_salu_ind, 16}, {"SALU/VALU alternating", k_mix_sv, 16}, {"s_nop 0", k_snop, 16},
                 {"dpp pair + s_nop + v_add (4 instr)", k_dpp_dep, 16}, {"readlane, s_nop 1, v_mul, s_nop 0 (4 instr)", k_readlane_use, 16},
                 {"v_cmp, s_and, v_cndmask, s_nop (4 instr)", k_cmp_sand, 16}, {"ds_read dependent chain (issue -> use)", k_lds_chain, 16},
                 {"ds_read issue (16 then wait)", k_lds_issue, 16}, {"global_load_dword issue (4 then wait, same line)", k_vmem_issue, 4},
                 {"global_load_dwordx4 issue (4 then wait)", k_vmem_issue4, 4}, {"global_load dependent chain (L1/L2 hit)", k_vmem_chain, 4},
                 {"s_cmp + s_cbranch not taken (2 instr)", k_branch_nt, 16}, {"s_branch taken over one instr", k_branch_tk, 16},
                 {"v_writelane m0", k_writelane, 16}, {"s_mul_i32 dependent", k_smul, 16}, {"s_ff1_i32_b64", k_ff1, 16},
                 {"[s_cmp, cbranch nt, 4 v_add] group of 6", k_br_d0, 4}, {"[s_cmp, 2 v_add, cbranch nt, 2 v_add] group of 6", k_br_d2, 4},
                 {"[s_cmp, 4 v_add, cbranch nt] group of 6", k_br_d4, 4}, {"[v_cmp, 4 v_add, cbranch_vccz nt] group of 6", k_br_vccz, 4},
                 {"[v_cmp, s_and, 4 v_add] group of 6", k_cmp_and_d0, 4}, {"[v_cmp, 2 v_add, s_and, 2 v_add] group of 6", k_cmp_and_d2, 4},
                 {"[v_cmp, 4 v_add, s_and] group of 6", k_cmp_and_d4, 4}, {"[s_and, v_cndmask, 4 v_add] group of 6", k_sand_cnd_d0, 4},
                 {"[readlane, s_add, 4 v_add] group of 6", k_rdl_salu_d0, 4}, {"[readlane, 4 v_add, s_add] group of 6", k_rdl_salu_d4, 4},
                 {"[s_add, v_add(sgpr), 4 v_add] group of 6", k_salu_valu_d0, 4},
                 {"global_load_dword, 16 lanes active", k_vmem_l16, 4}, {"global_load_dword, 32 lanes active", k_vmem_l32, 4},
                 {"global_load_dword, 48 lanes active", k_vmem_l48, 4}, {"[global_load_dword + 3 v_add] group", k_vmem_nowait, 4},
                 {"global_load_dwordx2", k_vmem_x2, 4}, {"s_load_dwordx4 issue", k_smem, 4}, {"s_load_dword chain (latency)", k_smem_chain, 4},
                 {"ds_read2_b32 issue", k_ds_read2, 16}, {"ds_write_b32 issue", k_ds_write, 16},
                 {"loop of 6 instr with taken back-edge, per iteration", k_loop_tk, 16}, {"v_add under s_setprio 3", k_setprio, 16},
                 {"16 global_load_dword, all lanes one address", k_vm16_same, 16}, {"16 global_load_dword, coalesced 256 B", k_vm16_coal, 16},
                 {"16 global_load_dword, 64 lanes 64 lines", k_vm16_scat, 16}, {"16 global_load_dword, 64 lanes over 8 lines", k_vm16_scat8, 16},
                 {"16 global_load_dwordx4, 64 lanes 64 lines (unaligned)", k_vm16_x4_scat, 16}, {"16 global_load_dwordx4, coalesced 1 KB", k_vm16_x4_coal, 16},
                 {"[global_load_dword scattered + 7 v_add] group of 8", k_vm_mix, 4}, {"16 s_load_dwordx8", k_sm16, 16}};
    for (auto &t : tests) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(64), 0, 0, d_out, d_sink, d_idx);
            hipDeviceSynchronize();
        }
        std::vector<Res> r(blocks);
        hipMemcpy(r.data(), d_out, sizeof(Res) * blocks, hipMemcpyDeviceToHost);
        double cyc = (double)r[0].cyc / ((double)ITER * t.per_iter), ns = (double)r[0].wall * 10.0 / ((double)ITER * t.per_iter);
        printf("%-52s %7.2f ticks %7.2f ns per unit  (s_memtime/wall = %.3f GHz)\n", t.name, cyc, ns, (double)r[0].cyc / ((double)r[0].wall * 10.0));
    }
    return 0;
}

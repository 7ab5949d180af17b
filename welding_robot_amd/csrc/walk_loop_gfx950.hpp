// walk_loop_gfx950.hpp -- the general step of the ant walk (ACS_Rank::selectNext, ACSRank_3D.hpp:134-193) as a
// hand-scheduled gfx950 assembly loop.  Included by acs_walk.hpp (needs WaRun, WaWalkState, wa_delta, wa_ctr_draw).
// Cost model it is scheduled for: tools/ubench/issue_rates.hip (profiles/r02/issue_rates.txt).
#pragma once

// ------------------------------------------------------------------ the hand-scheduled walk loop
// DEV mode, alpha == 1, dense field.  Same arithmetic, same operands, same order as wa_walk_fast (the tests hold the
// two against each other and against the oracle); what changes is what a LONE wavefront pays for: it issues one
// instruction per ~4 cycles whatever the type, so a step costs (instructions x 4 cycles) + whatever memory latency is
// left exposed.  The compiler's loop is ~85 instructions and waits for vmcnt(0) at its back edge; this one is ~60 and
// keeps three generations of loads in flight with exact vmcnt counts:
//   * records of the six neighbours (needed NEXT step) are requested first, then the records TWO hops away are touched
//     with two 16-byte loads nobody waits for: by the time they are requested for real they sit in L2 (vector memory
//     returns in order, so the younger touch loads never hold back the older record loads);
//   * lanes are laid out so that the roulette's first hit (scanning edge 5 down to 0) is the LOWEST set bit: role k
//     sits at position 5-k of its 8-lane block, `total` lands in position 0, one s_ff1 picks; block b holds the
//     record of neighbour 5-b, so the next active block is the picked lane's position;
//   * no address clamp (guard bands around the fields), no per-step L (table lookup at the end), the path word and the
//     new voxel id come out of ONE v_readlane, the tabu insert is an unconditional ds_write (other lanes hit a private
//     dummy slot), limits are checked per 64-step block instead of per step.
// Lane constants travel through LDS (inline asm takes at most 30 operands); temporaries are fixed registers.
// bytes behind the tabu hash: the sentinel slot (+ padding to 64 B), 64 dummy slots and two columns of 64 dwords that carry the lazy /
// rejoin variants' per-lane stamp offsets and scalars into the loop (the seven lane constants of every variant travel as operands; the
// diagnostic builds keep their six sums in further columns).  It counts: with a 2^12 table (pair planning) a walk block is 16 KB + this,
// and 160 KB of LDS hold NINE of them while this stays below 1 820 B (eight below 4 096 B)
#if defined(WA_ASM_STAMPS) || defined(WA_ASM_SPAN_A)
#define WA_WALK_LDS_EXTRA 4160
#define WA_LC_COL_STAMP 13
#define WA_LC_COL_PARAM 14
#define WA_LC_OFF_STAMP "3328"
#define WA_LC_OFF_PARAM "3584"
#else
#define WA_WALK_LDS_EXTRA 832    // 64 + 256 + 2 x 256
#define WA_LC_COL_STAMP 0
#define WA_LC_COL_PARAM 1
#define WA_LC_OFF_STAMP "0"
#define WA_LC_OFF_PARAM "256"
#endif
#define WA_WALK_LDS_PAD 16       // entries between the table and the dummy slots; entry 0 of them is the sentinel (never empty, never a key)
#define WA_ASM_DPP_C " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"   // lane i <- lane i-1: prob_sum grows from role 5 (position 0) upwards
#define WA_ASM_DPP_T " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"   // lane i <- lane i+1: total flows down into position 0
// The records two hops away are touched with two 16-byte loads nobody waits for (W = SELF), or not at all (W = NONE):
//   SELF pays when ONE search owns the GPU (a walk block per CU, a lone wave per SIMD): the touches put the next records into the
//        CU's L1 / the XCD's L2 before the real loads ask for them (generation 0-10 walk launches at C3: 186 vs 200 us);
//   NONE pays when many searches saturate it (BASELINE config C5: 2 304 walk blocks): the touches are two thirds of the cache
//        lines a step requests, the walk is then bound by L2 / address-pipeline throughput, not by latency (C5: 0.99 -> 0.87 s).
#define WA_ASM_WARM_ADDR_SELF "v_add_u32 v83, s40, v66\n"
#define WA_ASM_WARM0_SELF "global_load_dwordx4 v[86:89], v83, %[pher]\n"
#define WA_ASM_WARM1_SELF "global_load_dwordx4 v[90:93], v83, %[heur]\n"
// -DWA_EXP_STORE=1 (experiment, tools/walk_ab.py; round 5): what would a per-step global STORE cost the lone search's step -- the insert of a
// visited bitmap kept in device memory instead of the LDS hash (task 5)?  Vector memory operations retire in order and the
// loop's waits are exact vmcnt counts: a store issued in step t must have been acknowledged before the records requested behind it can
// count as arrived (step t + 2).  The variant stores the step's path register over the path block that is being collected (rewritten in
// full when the block completes: harmless) -- four address instructions + one global_store_dword, vmcnt counts raised by one;
// -DWA_EXP_STORE=2: the four instructions without the store.  Dense loop with touch loads only.
#if defined(WA_EXP_STORE)
#define WA_ASM_EXP_ADDR "s_and_b32 s46, m0, 0xffffffc0\n" "s_lshl_b32 s46, s46, 2\n" "v_lshlrev_b32 v94, 2, v64\n" "v_add_u32 v94, s46, v94\n"
#if WA_EXP_STORE == 1
#define WA_ASM_EXP_STORE WA_ASM_EXP_ADDR "global_store_dword v94, %[pbuf], %[path]\n"
#define WA_ASM_VMWAIT_SELF "s_waitcnt vmcnt(5)\n"
#else
#define WA_ASM_EXP_STORE WA_ASM_EXP_ADDR
#define WA_ASM_VMWAIT_SELF "s_waitcnt vmcnt(4)\n"
#endif
#else
#define WA_ASM_EXP_STORE ""
#define WA_ASM_VMWAIT_SELF "s_waitcnt vmcnt(4)\n"
#endif
#define WA_ASM_VMWAIT_LAZY_SELF "s_waitcnt vmcnt(5)\n"          /* (one more load per step: the stamp) */
#define WA_ASM_WARM_ADDR_NONE ""
#define WA_ASM_WARM0_NONE "s_nop 0\n"                           /* (keeps the compares four instructions away from their scalar consumers) */
#define WA_ASM_WARM1_NONE "s_nop 0\n"
#define WA_ASM_VMWAIT_NONE "s_waitcnt vmcnt(2)\n"
#define WA_ASM_VMWAIT_LAZY_NONE "s_waitcnt vmcnt(3)\n"
// W = DIRECT (round 5): NO look-ahead.  The step after the pick requests only the records of the voxel the ant has just moved to
// (one 24-byte record per field instead of six: 2-4 cache lines named per step instead of 12-14) and the next step waits for them.
// What a saturated launch pays for the look-ahead is memory-system load (profiles/r04/pmc_walk_p8.txt: 4.3 L1->L2 requests and ~2 L2
// misses per step for the two records a step uses); what DIRECT pays is the load's latency on every step's chain, which the other
// resident wavefronts of a saturated launch are there to cover.  Same lanes, same arithmetic: every lane block requests the record of
// `cur` itself (lane constant dj = 0, see wa_walk_fast_asm), the active block is always block 0.
#define WA_ASM_WARM_ADDR_DIRECT ""
#define WA_ASM_WARM0_DIRECT "s_nop 0\n"
#define WA_ASM_WARM1_DIRECT "s_nop 0\n"
#define WA_ASM_VMWAIT_DIRECT "s_waitcnt vmcnt(0)\n"
#define WA_ASM_VMWAIT_LAZY_DIRECT "s_waitcnt vmcnt(0)\n"
// where the step's tail puts the records it requests: look-ahead -> where THIS step's records were (needed by the step after the next
// one); DIRECT -> the other register set (needed by the NEXT step), loads in front of the probe (they are what the next step waits for)
#define WA_ASM_REQ_SELF(CP, CH, CS, NP, NH, NS, HEAD, T) WA_ASM_NEXT(CP, CH, CS, HEAD, T)
#define WA_ASM_REQ_NONE(CP, CH, CS, NP, NH, NS, HEAD, T) WA_ASM_NEXT(CP, CH, CS, HEAD, T)
#define WA_ASM_REQ_DIRECT(CP, CH, CS, NP, NH, NS, HEAD, T) WA_ASM_NEXT_LOADS(NP, NH, NS, HEAD) WA_SPAN_8 WA_ASM_NEXT_PROBE_##T
// the block that holds the next step's records: the picked lane's position -- or, DIRECT, block 0 for ever (s[54:55] stays 63 << 0
// until an event zeroes it; the event handler restores it from %[g8] = 0)
#define WA_ASM_ACTIVE_SELF "s_lshl_b32 %[g8], s45, 3\n" "s_lshl_b64 s[54:55], 63, %[g8]\n"
#define WA_ASM_ACTIVE_NONE WA_ASM_ACTIVE_SELF
#define WA_ASM_ACTIVE_DIRECT ""
// -DWA_ASM_STAMPS (diagnostic builds, tools/walk_stamps_asm.py): s_memtime at six points of the step, differences summed in
// s72..s77 (s70 = previous stamp); each stamp drains LDS and costs ~40 cycles: read the shares, not the totals
// -DWA_ASM_SPAN_A=a -DWA_ASM_SPAN_B=b (diagnostic builds, tools/walk_spans.py): ONE pair of s_memtime per step, at points a and b of
// the step (0 top, 1 after the LDS wait, 2 after the record wait, 3 before the masks, 4 before the sums, 5 after them, 6 after the
// rare-event branch, 7 after the insert, 8 between record loads and probe, 9 after both, 10 end), nobody waits for them: the step's own
// lgkmcnt(0) at the head of the NEXT step covers them, and the difference is summed there (s72).  a == b: the time between two
// consecutive passes of that point = the whole step.  Two instructions per step instead of six draining stamps.
#if defined(WA_ASM_SPAN_A)
#define WA_ASM_STAMPS 1
#define WA_ASM_STAMP(i) ""
#define WA_ASM_SPAN_ACC "s_sub_u32 s64, s62, s60\n s_cmp_eq_u32 s60, 0\n s_cselect_b32 s64, 0, s64\n s_cmp_eq_u32 s62, 0\n s_cselect_b32 s64, 0, s64\n s_add_u32 s72, s72, s64\n" WA_ASM_SPAN_RESET
#if WA_ASM_SPAN_A == WA_ASM_SPAN_B
#define WA_ASM_SPAN_RESET ""
#define WA_ASM_SPAN_AT "s_mov_b32 s60, s62\n s_memtime s[62:63]\n"
#else
#define WA_ASM_SPAN_RESET "s_mov_b32 s60, 0\n s_mov_b32 s62, 0\n"
#endif
#if WA_ASM_SPAN_A == 0 && WA_ASM_SPAN_B == 0
#define WA_SPAN_0 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 0
#define WA_SPAN_0 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 0
#define WA_SPAN_0 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_0 ""
#endif
#if WA_ASM_SPAN_A == 1 && WA_ASM_SPAN_B == 1
#define WA_SPAN_1 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 1
#define WA_SPAN_1 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 1
#define WA_SPAN_1 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_1 ""
#endif
#if WA_ASM_SPAN_A == 2 && WA_ASM_SPAN_B == 2
#define WA_SPAN_2 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 2
#define WA_SPAN_2 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 2
#define WA_SPAN_2 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_2 ""
#endif
#if WA_ASM_SPAN_A == 3 && WA_ASM_SPAN_B == 3
#define WA_SPAN_3 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 3
#define WA_SPAN_3 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 3
#define WA_SPAN_3 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_3 ""
#endif
#if WA_ASM_SPAN_A == 4 && WA_ASM_SPAN_B == 4
#define WA_SPAN_4 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 4
#define WA_SPAN_4 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 4
#define WA_SPAN_4 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_4 ""
#endif
#if WA_ASM_SPAN_A == 5 && WA_ASM_SPAN_B == 5
#define WA_SPAN_5 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 5
#define WA_SPAN_5 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 5
#define WA_SPAN_5 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_5 ""
#endif
#if WA_ASM_SPAN_A == 6 && WA_ASM_SPAN_B == 6
#define WA_SPAN_6 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 6
#define WA_SPAN_6 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 6
#define WA_SPAN_6 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_6 ""
#endif
#if WA_ASM_SPAN_A == 7 && WA_ASM_SPAN_B == 7
#define WA_SPAN_7 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 7
#define WA_SPAN_7 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 7
#define WA_SPAN_7 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_7 ""
#endif
#if WA_ASM_SPAN_A == 8 && WA_ASM_SPAN_B == 8
#define WA_SPAN_8 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 8
#define WA_SPAN_8 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 8
#define WA_SPAN_8 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_8 ""
#endif
#if WA_ASM_SPAN_A == 9 && WA_ASM_SPAN_B == 9
#define WA_SPAN_9 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 9
#define WA_SPAN_9 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 9
#define WA_SPAN_9 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_9 ""
#endif
#if WA_ASM_SPAN_A == 10 && WA_ASM_SPAN_B == 10
#define WA_SPAN_10 WA_ASM_SPAN_AT
#elif WA_ASM_SPAN_A == 10
#define WA_SPAN_10 "s_memtime s[60:61]\n"
#elif WA_ASM_SPAN_B == 10
#define WA_SPAN_10 "s_memtime s[62:63]\n"
#else
#define WA_SPAN_10 ""
#endif
#define WA_ASM_COUNT_COLL "s_add_u32 s73, s73, 1\n"
#define WA_ASM_COUNT_EVENT "s_add_u32 s74, s74, 1\n"
#else
#define WA_ASM_SPAN_ACC ""
#define WA_ASM_COUNT_COLL ""
#define WA_ASM_COUNT_EVENT ""
#define WA_SPAN_0 ""
#define WA_SPAN_1 ""
#define WA_SPAN_2 ""
#define WA_SPAN_3 ""
#define WA_SPAN_4 ""
#define WA_SPAN_5 ""
#define WA_SPAN_6 ""
#define WA_SPAN_7 ""
#define WA_SPAN_8 ""
#define WA_SPAN_9 ""
#define WA_SPAN_10 ""
#endif
#if defined(WA_ASM_SPAN_A)
#elif defined(WA_ASM_STAMPS)
#define WA_ASM_STAMP(i) "s_memtime s[60:61]\n" "s_waitcnt lgkmcnt(0)\n" "s_sub_u32 s62, s60, s70\n" "s_add_u32 s" #i ", s" #i ", s62\n" "s_mov_b32 s70, s60\n"
#else
#define WA_ASM_STAMP(i) ""
#endif
// One step.  CP/CH = records of `cur` (requested by the step before the previous one's tail, arrived or arriving); the
// records of the NEXT voxel's neighbours are requested into the same registers as soon as that voxel is known (the
// step in between runs on the other register set NP/NH).  X = label suffix.
// Schedule rules (measured with tools/ubench/issue_rates.hip, one wave alone on its SIMD): every instruction costs
// ~4.3 cycles; an SALU instruction reading an SGPR/VCC that a VALU instruction wrote stalls ~16 cycles unless four
// other instructions sit between them; a conditional branch costs ~13 cycles not taken, ~23 taken.  Hence: compares
// early, their scalar consumers late, and ONE rare-event branch per step -- a completed 64-word block and the arrival
// zero the active-lane mask, so the NEXT step finds no candidate and leaves through the same exit as a dead end.
// dense field: info = |p| * h.  lazy field (wa_acs_create_lazy): the stamp of cur's record decides -- 0: never deposited, every
// admissible edge is worth the clean value; else the stored value with the evaporations it has missed applied one by one
#define WA_ASM_INFO_DENSE(CP, CH, CS, X) "v_mul_f32 v78, |" CP "|, " CH "\n"              /* info (:154), alpha == 1 */
#define WA_ASM_INFO_LAZY(CP, CH, CS, X)                                                                           \
    "v_readlane_b32 s36, " CS ", %[g8]\n"                         /* stamp of cur: the same in the six lanes of the active block */ \
    "v_mul_f32 v78, s33, " CH "\n"                                /* clean value x heuristic (the common case in pair planning) */
#define WA_ASM_INFO2_DENSE(X) ""
#define WA_ASM_INFO2_LAZY(X)                                                                                      \
    "s_cmp_eq_u32 s36, 0\n"                                                                                       \
    "s_cbranch_scc0 Lwa_dirty_" X "%=\n"                                                                          \
    "Lwa_info_" X "%=:\n"
#define WA_ASM_HEAD_DENSE(NS) ""
#define WA_ASM_HEAD_LAZY(NS)                                                                                      \
    "s_lshl_b32 s46, %[cur], 2\n"                                                                                 \
    "v_add_u32 v98, s46, v99\n"                                                                                   \
    "global_load_dword " NS ", v98, %[stamp]\n"                   /* the neighbour's stamp travels with its record */
// a deposited record: stored value -> value after the evaporations it missed (one rounding per multiplication, like the sweep)
#define WA_ASM_DIRTY(CP, CH, X)                                                                                   \
    "Lwa_dirty_" X "%=:\n"                                                                                        \
    "s_sub_u32 s37, s34, s36\n"                                   /* evap_now + 1 - stamp */                      \
    "v_and_b32 v78, 0x7fffffff, " CP "\n"                                                                         \
    "s_cmp_eq_u32 s37, 0\n"                                                                                       \
    "s_cbranch_scc1 Lwa_dirty2_" X "%=\n"                                                                         \
    "Lwa_catch4_" X "%=:\n"                                       /* four pending evaporations at a time (one rounding each, in order) ... */ \
    "s_cmp_lt_u32 s37, 4\n"                                                                                       \
    "s_cbranch_scc1 Lwa_catch1_" X "%=\n"                                                                         \
    "v_mul_f32 v78, s35, v78\n"                                                                                   \
    "v_mul_f32 v78, s35, v78\n"                                                                                   \
    "v_mul_f32 v78, s35, v78\n"                                                                                   \
    "v_mul_f32 v78, s35, v78\n"                                                                                   \
    "s_sub_u32 s37, s37, 4\n"                                                                                     \
    "s_branch Lwa_catch4_" X "%=\n"                                                                               \
    "Lwa_catch1_" X "%=:\n"                                       /* ... then the last one to three */             \
    "s_cmp_eq_u32 s37, 0\n"                                                                                       \
    "s_cbranch_scc1 Lwa_dirty2_" X "%=\n"                                                                         \
    "Lwa_catch_" X "%=:\n"                                                                                        \
    "v_mul_f32 v78, s35, v78\n"                                                                                   \
    "s_sub_u32 s37, s37, 1\n"                                                                                     \
    "s_cmp_lg_u32 s37, 0\n"                                                                                       \
    "s_cbranch_scc1 Lwa_catch_" X "%=\n"                                                                          \
    "Lwa_dirty2_" X "%=:\n"                                                                                       \
    "v_mul_f32 v78, v78, " CH "\n"                                                                                \
    "s_branch Lwa_info_" X "%=\n"
// rejoin watch (ants that left the best path after replaying a prefix of it, wa_walk_one): the membership stamp of the voxel
// the ant stands on is fetched with a scalar load and looked at one step later -- when it says "on the best path" (and the
// hold-off counter has run out) the loop hands back to the caller, who tries to put the ant back on the replay track
#define WA_ASM_REJ_NONE(X) ""
#define WA_ASM_REJ_WATCH(X)                                                                                       \
    "s_sub_u32 s81, s81, 1\n"                                     /* steps left before a rejoin may be reported */ \
    "s_cmp_eq_u32 s78, s80\n"                                     /* the voxel we stood on one step ago: on the best path? */ \
    "s_cselect_b32 s82, s81, 1\n"                                                                                 \
    "s_cmp_le_i32 s82, 0\n"                                                                                       \
    "s_cbranch_scc1 Lwa_rej_" X "%=\n"                                                                            \
    "s_lshl_b32 s83, %[cur], 2\n"                                                                                 \
    "s_load_dword s78, %[markb], s83\n"
#define WA_ASM_REJ_EXIT(CP, CH, X)                                                                                \
    "Lwa_rej_" X "%=:\n"                                                                                          \
    "v_mov_b32 %[pio], " CP "\n"                                                                                  \
    "v_mov_b32 %[hio], " CH "\n"                                                                                  \
    "s_mov_b32 %[code], 4\n"                                                                                      \
    "s_branch Lwa_out%=\n"
#define WA_ASM_MASKS(X)                                                                                           \
    "s_and_b64 s[48:49], s[48:49], vcc\n"                                                                         \
    "s_cbranch_scc1 Lwa_coll_" X "%=\n"                                                                           \
    "s_and_b64 s[50:51], s[50:51], vcc\n"                         /* ... and not visited (:145) */                \
    "s_and_b64 s[52:53], s[50:51], s[54:55]\n"                    /* ... in the active block */
// the two ordered fp32 sums (:170 total, ascending k; :178 prob_sum, descending k) as 5 + 5 DPP row shifts; a DPP read of a
// register written by the previous VALU instruction needs two wait states: the other chain's add + one s_nop
#define WA_ASM_SUMS                                                                                               \
    "v_add_f32_dpp v80, v78, v78" WA_ASM_DPP_C                                                                    \
    "v_add_f32_dpp v79, v78, v78" WA_ASM_DPP_T                                                                    \
    "s_nop 0\n"                                                                                                   \
    "v_add_f32_dpp v80, v80, v78" WA_ASM_DPP_C                                                                    \
    "v_add_f32_dpp v79, v79, v78" WA_ASM_DPP_T                                                                    \
    "s_nop 0\n"                                                                                                   \
    "v_add_f32_dpp v80, v80, v78" WA_ASM_DPP_C                                                                    \
    "v_add_f32_dpp v79, v79, v78" WA_ASM_DPP_T                                                                    \
    "s_nop 0\n"                                                                                                   \
    "v_add_f32_dpp v80, v80, v78" WA_ASM_DPP_C                                                                    \
    "v_add_f32_dpp v79, v79, v78" WA_ASM_DPP_T                                                                    \
    "s_nop 0\n"                                                                                                   \
    "v_add_f32_dpp v80, v80, v78" WA_ASM_DPP_C                                                                    \
    "v_add_f32_dpp v79, v79, v78" WA_ASM_DPP_T
#define WA_ASM_NEXT_LOADS(CP, CH, CS, HEAD)                                                                       \
    "s_mul_i32 s40, %[cur], 24\n"                                 /* records of the new voxel's six neighbours: needed by the step after */ \
    "v_add_u32 v82, s40, v65\n"                                   /* the next one; they go where this step's records were */ \
    "global_load_dword " CP ", v82, %[pher]\n"                                                                    \
    "global_load_dword " CH ", v82, %[heur]\n"                                                                    \
    HEAD(CS)
// 16-bit entries (T16, see WaTabu in acs_walk.hpp; saturated launches): t = id * kmul holds the bijective hash in its top bits; %[hs] is 31 - hl and
// %[hm4] (size - 1) * 2 there -- the slot's BYTE address --, s68 = kmul, s69 = 16 - hl, s71 = 0xffff (prologue); v76 = what the slot holds if it holds
// this lane's key at the displacement the lane's chain stands on (the collision path adds one per slot)
#define WA_ASM_NEXT_PROBE_T16                                                                                     \
    "s_mul_i32 s41, %[cur], s68\n"                                                                                \
    "v_add_u32 v75, s41, v67\n"                                                                                   \
    "v_lshrrev_b32 v77, %[hs], v75\n"                                                                             \
    "v_and_b32 v77, %[hm4], v77\n"                                                                                \
    "ds_read_u16 v100, v77\n"                                      /* next step's tabu probe: the slot ... */       \
    "ds_read_u16 v101, v77 offset:2\n"                             /* ... and its successor (the sentinel behind the last slot) */ \
    "v_lshrrev_b32 v76, s69, v75\n"                                                                               \
    "v_and_b32 v76, 0xfff0, v76\n"
#define WA_ASM_NEXT_PROBE_T32 WA_ASM_NEXT_PROBE
#define WA_ASM_CMPS_T32                                                                                           \
    "v_cmp_ne_u32 vcc, v100, v76\n"                               /* probed slot does not hold the neighbour */   \
    "v_cmp_ne_u32 s[48:49], -1, v100\n"                           /* ... and is not empty: chain goes on */
#define WA_ASM_CMPS_T16                                                                                           \
    "v_cmp_ne_u32 vcc, v100, v76\n"                                                                               \
    "v_cmp_ne_u32 s[48:49], s71, v100\n"
#define WA_ASM_INSERT_T32 "ds_write_b32 v84, v76\n"
#define WA_ASM_INSERT_T16 "ds_write_b16 v84, v76\n"
#define WA_ASM_NEXT_PROBE                                                                                         \
    "s_mul_i32 s41, %[cur], 0x9e3779b1\n"                                                                         \
    "v_add_u32 v77, s41, v67\n"                                                                                   \
    "v_lshrrev_b32 v77, %[hs], v77\n"                                                                             \
    "v_lshlrev_b32 v77, 2, v77\n"                                                                                 \
    "ds_read2_b32 v[100:101], v77 offset1:1\n"                     /* next step's tabu probe: the slot and its successor (the chain's second slot) */ \
    "v_add_u32 v76, %[cur], v68\n"
// the probe goes first: its LDS round trip (~50 cycles) then runs under the record loads' issue instead of in front of the next
// step's head (measured, tools/walk_ab.py: -1 % on the step)
#define WA_ASM_NEXT(CP, CH, CS, HEAD, T) WA_ASM_NEXT_PROBE_##T WA_SPAN_8 WA_ASM_NEXT_LOADS(CP, CH, CS, HEAD)
#define WA_ASM_STEP(CP, CH, NP, NH, X, W, T) WA_ASM_STEP_G(CP, CH, NP, NH, "", "", X, WA_ASM_HEAD_DENSE, WA_ASM_INFO_DENSE, WA_ASM_INFO2_DENSE, WA_ASM_VMWAIT_##W, WA_ASM_REJ_NONE, W, T)
#define WA_ASM_STEP_REJ(CP, CH, NP, NH, X, W, T) WA_ASM_STEP_G(CP, CH, NP, NH, "", "", X, WA_ASM_HEAD_DENSE, WA_ASM_INFO_DENSE, WA_ASM_INFO2_DENSE, WA_ASM_VMWAIT_##W, WA_ASM_REJ_WATCH, W, T)
#define WA_ASM_STEP_LAZY(CP, CH, CS, NP, NH, NS, X, W, T) WA_ASM_STEP_G(CP, CH, NP, NH, CS, NS, X, WA_ASM_HEAD_LAZY, WA_ASM_INFO_LAZY, WA_ASM_INFO2_LAZY, WA_ASM_VMWAIT_LAZY_##W, WA_ASM_REJ_NONE, W, T)
#define WA_ASM_STEP_LAZY_REJ(CP, CH, CS, NP, NH, NS, X, W, T) WA_ASM_STEP_G(CP, CH, NP, NH, CS, NS, X, WA_ASM_HEAD_LAZY, WA_ASM_INFO_LAZY, WA_ASM_INFO2_LAZY, WA_ASM_VMWAIT_LAZY_##W, WA_ASM_REJ_WATCH, W, T)
#define WA_ASM_STEP_G(CP, CH, NP, NH, CS, NS, X, HEAD, INFO, INFO2, VMWAIT, REJ, W, T)                             \
    WA_SPAN_0                                                                                                     \
    WA_ASM_WARM_ADDR_##W                                          /* (s40 = cur * 24 since the previous step's tail) */ \
    WA_ASM_STAMP(72)                                                                                              \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    WA_ASM_SPAN_ACC                                                                                               \
    WA_SPAN_1                                                                                                     \
    WA_ASM_STAMP(73)                                                                                              \
    REJ(X)                                                        /* once per step: a re-evaluation (collision, block boundary) enters below */ \
    "Lwa_redo_" X "%=:\n"                                                                                         \
    WA_ASM_CMPS_##T                                                                                               \
    VMWAIT                                                        /* records of cur; the touch loads + the new ones stay in flight */ \
    WA_SPAN_2                                                                                                     \
    WA_ASM_STAMP(74)                                                                                              \
    "v_cmp_lt_i32 s[50:51], -1, " CP "\n"                         /* sign clear: in bounds and free (:148) */     \
    INFO(CP, CH, CS, X)                                                                                           \
    WA_ASM_WARM0_##W                                              /* records two hops away: touched, never waited for */ \
    WA_ASM_WARM1_##W                                                                                              \
    INFO2(X)                                                                                                      \
    WA_SPAN_3                                                                                                     \
    "Lwa_masks_" X "%=:\n"                                        /* a collision comes back here: records, info and touch loads are done */ \
    WA_ASM_MASKS(X)                                                                                               \
    "v_cndmask_b32 v78, 0, v78, s[52:53]\n"                                                                       \
    WA_SPAN_4                                                                                                     \
    "v_readlane_b32 s42, %[ub], m0\n"                             /* this step's uniform draw (:169) */           \
    "v_add_u32 v85, %[cur], v69\n"                                /* candidate path words: (cur + d_k) | k << 29 */ \
    WA_ASM_SUMS                                                                                                   \
    WA_SPAN_5                                                                                                     \
    WA_ASM_STAMP(75)                                                                                              \
    "v_mul_f32 v81, s42, v79\n"                                   /* rnd = u * total (:170), valid in position 0 */ \
    "s_nop 0\n"                                                                                                   \
    "v_readlane_b32 s43, v81, %[g8]\n"                                                                            \
    "s_nop 1\n"                                                                                                   \
    "v_cmp_le_f32 vcc, s43, v80\n"                                /* prob_sum >= rnd (:178) */                    \
    "s_and_b64 s[56:57], vcc, s[52:53]\n"                                                                         \
    "s_cbranch_scc0 Lwa_rare_" X "%=\n"                           /* no candidate (:162), fall-through (:191), or a pending event */ \
    WA_SPAN_6                                                                                                     \
    WA_ASM_STAMP(76)                                                                                              \
    "s_ff1_i32_b64 s45, s[56:57]\n"                               /* first hit scanning edge 5 -> 0 */            \
    "v_cmp_eq_u32 vcc, s45, v64\n"                                                                                \
    "v_readlane_b32 s44, v85, s45\n"                              /* path word of the move */                     \
    "v_cndmask_b32 v84, v70, v77, vcc\n"                                                                          \
    WA_ASM_INSERT_##T                                             /* addNextNode :75 -- the probe ended on the free slot */ \
    WA_ASM_EXP_STORE                                                                                              \
    WA_SPAN_7                                                                                                     \
    WA_ASM_ACTIVE_##W                                             /* next active block = position of the pick (low 6 bits count) */ \
    "s_and_b32 %[cur], s44, 0x1fffffff\n"                                                                         \
    WA_ASM_REQ_##W(CP, CH, CS, NP, NH, NS, HEAD, T)                                                              \
    WA_SPAN_9                                                                                                     \
    "v_writelane_b32 %[pbuf], s44, m0\n"                          /* :76-77 */                                    \
    "s_add_i32 m0, m0, 1\n"                                                                                       \
    "s_and_b32 s46, m0, %[em]\n"                                  /* 63, or 15 once the straggler check has seen arrivals */ \
    "s_cselect_b64 s[54:55], s[54:55], 0\n"                       /* block of 64 path words complete (or a check is due): event */ \
    "s_cmp_lg_u32 %[cur], %[end]\n"                                                                               \
    "s_cselect_b64 s[54:55], s[54:55], 0\n"                       /* arrived (:182): event */                     \
    WA_SPAN_10                                                                                                    \
    WA_ASM_STAMP(77)
// some lane's probe hit another key: advance those lanes along their chains, compare again and re-enter at the masks
#define WA_ASM_COLL(X, T) WA_ASM_COLL_##T(X)
// the same for 16-bit entries: two bytes per slot, the lane's compare value follows its chain (displacement + 1 per slot); a chain that would stand 14
// slots behind its home slot cannot be decided by an entry any more -> Lwa_ovf: the loop hands back with code 6 and the walk spills to its bitmap
#define WA_ASM_COLL_T16(X)                                                                                        \
    "Lwa_coll_" X "%=:\n"                                                                                         \
    WA_ASM_COUNT_COLL                                                                                             \
    "s_mov_b64 s[58:59], exec\n"                                                                                  \
    "s_mov_b64 exec, s[48:49]\n"                                                                                  \
    "v_mov_b32 v100, v101\n"                                                                                      \
    "v_add_u32 v77, 2, v77\n"                                                                                     \
    "v_and_b32 v77, %[hm4], v77\n"                                                                                \
    "v_add_u32 v76, 1, v76\n"                                                                                     \
    "s_mov_b64 exec, s[58:59]\n"                                                                                  \
    "Lwa_coll_cmp_" X "%=:\n"                                                                                     \
    "v_cmp_ne_u32 vcc, v100, v76\n"                                                                               \
    "v_cmp_ne_u32 s[48:49], s71, v100\n"                                                                          \
    "s_nop 3\n"                                                                                                   \
    "s_and_b64 s[48:49], s[48:49], vcc\n"                                                                         \
    "s_cbranch_scc0 Lwa_masks_" X "%=\n"                                                                          \
    "s_mov_b64 exec, s[48:49]\n"                                                                                  \
    "ds_read_u16 v100, v77\n"                                                                                     \
    "s_mov_b64 exec, s[58:59]\n"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "v_cmp_ne_u32 vcc, v100, v76\n"                                                                               \
    "v_cmp_ne_u32 s[48:49], s71, v100\n"                                                                          \
    "s_nop 3\n"                                                                                                   \
    "s_and_b64 s[48:49], s[48:49], vcc\n"                                                                         \
    "s_cbranch_scc0 Lwa_masks_" X "%=\n"                                                                          \
    "s_mov_b64 exec, s[48:49]\n"                                                                                  \
    "v_add_u32 v77, 2, v77\n"                                                                                     \
    "v_and_b32 v77, %[hm4], v77\n"                                                                                \
    "v_add_u32 v76, 1, v76\n"                                                                                     \
    "v_and_b32 v94, 15, v76\n"                                    /* the displacement the chain stands on now */   \
    "v_cmp_lt_u32 vcc, 13, v94\n"                                                                                 \
    "ds_read_u16 v100, v77\n"                                                                                     \
    "s_mov_b64 exec, s[58:59]\n"                                                                                  \
    "s_nop 3\n"                                                                                                   \
    "s_cbranch_vccnz Lwa_ovf%=\n"                                                                                 \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "s_branch Lwa_coll_cmp_" X "%=\n"
#define WA_ASM_COLL_T32(X)                                                                                        \
    "Lwa_coll_" X "%=:\n"                                                                                         \
    WA_ASM_COUNT_COLL                                                                                             \
    "s_mov_b64 s[58:59], exec\n"                                                                                  \
    "s_mov_b64 exec, s[48:49]\n"                                                                                  \
    "v_mov_b32 v100, v101\n"                                      /* the chain's second slot came with the probe: no LDS round trip */ \
    "v_add_u32 v77, 4, v77\n"                                                                                     \
    "v_and_b32 v77, %[hm4], v77\n"                                /* (a probe of the table's LAST slot read the sentinel as its successor: it compares */ \
    "s_mov_b64 exec, s[58:59]\n"                                  /*  as another key and the loop below reads the real successor, slot 0) */ \
    "Lwa_coll_cmp_" X "%=:\n"                                                                                     \
    "v_cmp_ne_u32 vcc, v100, v76\n"                               /* the two probe compares again; sign mask and info of the first pass stand */ \
    "v_cmp_ne_u32 s[48:49], -1, v100\n"                                                                           \
    "s_nop 3\n"                                                                                                   \
    "s_and_b64 s[48:49], s[48:49], vcc\n"                                                                         \
    "s_cbranch_scc0 Lwa_masks_" X "%=\n"                                                                          \
    "s_mov_b64 exec, s[48:49]\n"                                  /* still another key: read the slot the chain stands on (after a fast pass the one */ \
    "ds_read_b32 v100, v77\n"                                     /* just compared, else the next), and advance behind it for the pass after */ \
    "s_mov_b64 exec, s[58:59]\n"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "v_cmp_ne_u32 vcc, v100, v76\n"                                                                               \
    "v_cmp_ne_u32 s[48:49], -1, v100\n"                                                                           \
    "s_nop 3\n"                                                                                                   \
    "s_and_b64 s[48:49], s[48:49], vcc\n"                                                                         \
    "s_cbranch_scc0 Lwa_masks_" X "%=\n"                                                                          \
    "s_mov_b64 exec, s[48:49]\n"                                                                                  \
    "v_add_u32 v77, 4, v77\n"                                                                                     \
    "v_and_b32 v77, %[hm4], v77\n"                                                                                \
    "ds_read_b32 v100, v77\n"                                                                                     \
    "s_mov_b64 exec, s[58:59]\n"                                                                                  \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "s_branch Lwa_coll_cmp_" X "%=\n"
// no candidate: a pending event (block complete / arrived: active mask zeroed by the previous step) or a dead end.
// CP/CH are the records of `cur`: they go back to the caller, who re-enters after a block boundary.
#define WA_ASM_RARE_BODY(CP, CH, IDX)                                                                             \
    "v_mov_b32 %[pio], " CP "\n"                                                                                  \
    "v_mov_b32 %[hio], " CH "\n"                                                                                  \
    "s_mov_b32 s47, " IDX "\n"                                                                                    \
    "s_cmp_eq_u64 s[54:55], 0\n"                                                                                  \
    "s_cbranch_scc0 Lwa_dead%=\n"                                                                                 \
    "s_branch Lwa_event%=\n"
#define WA_ASM_RARE(CP, CH, X, IDX) "Lwa_rare_" X "%=:\n" WA_ASM_RARE_BODY(CP, CH, IDX)

// the pieces of the loop's assembly text that the dense and the lazy variant share
#define WA_ASM_PROLOGUE_T32                                                                                           \
    "v_mov_b32 v64, %[c0]\n"                     /* the seven lane constants: lane, record offset, touch offset, hash term, */ \
    "v_mov_b32 v65, %[c1]\n"                     /* neighbour offset, path-word term, dummy slot (see wa_walk_fast_asm) */      \
    "v_mov_b32 v66, %[c2]\n"                                                                                      \
    "v_mov_b32 v67, %[c3]\n"                                                                                      \
    "v_mov_b32 v68, %[c4]\n"                                                                                      \
    "v_mov_b32 v69, %[c5]\n"                                                                                      \
    "v_mov_b32 v70, %[c6]\n"                                                                                      \
    "v_mov_b32 v71, %[pio]\n"                                                                                     \
    "v_mov_b32 v72, %[hio]\n"                                                                                     \
    "s_lshl_b64 s[54:55], 63, %[g8]\n"                                                                            \
    "s_mov_b32 m0, %[len]\n"                      /* the node count lives in m0: lane select of the draw and of the path word */ \
    "s_mul_i32 s41, %[cur], 0x9e3779b1\n"                                                                         \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "v_add_u32 v77, s41, v67\n"                                                                                   \
    "v_lshrrev_b32 v77, %[hs], v77\n"                                                                             \
    "v_lshlrev_b32 v77, 2, v77\n"                                                                                 \
    "ds_read2_b32 v[100:101], v77 offset1:1\n"                                                                    \
    "v_add_u32 v76, %[cur], v68\n"                                                                                \
    "s_mul_i32 s40, %[cur], 24\n"                /* records of cur's six neighbours: what the first step's tail would have requested */ \
    "v_add_u32 v82, s40, v65\n"                                                                                   \
    "global_load_dword v73, v82, %[pher]\n"                                                                       \
    "global_load_dword v74, v82, %[heur]\n"
#define WA_ASM_PROLOGUE_T16                                                                                           \
    "v_mov_b32 v64, %[c0]\n"                     /* the seven lane constants: lane, record offset, touch offset, hash term, */ \
    "v_mov_b32 v65, %[c1]\n"                     /* neighbour offset, path-word term, dummy slot (see wa_walk_fast_asm) */      \
    "v_mov_b32 v66, %[c2]\n"                                                                                      \
    "v_mov_b32 v67, %[c3]\n"                                                                                      \
    "v_mov_b32 v68, %[c4]\n"                                                                                      \
    "v_mov_b32 v69, %[c5]\n"                                                                                      \
    "v_mov_b32 v70, %[c6]\n"                                                                                      \
    "v_mov_b32 v71, %[pio]\n"                                                                                     \
    "v_mov_b32 v72, %[hio]\n"                                                                                     \
    "s_lshl_b64 s[54:55], 63, %[g8]\n"                                                                            \
    "s_mov_b32 m0, %[len]\n"                      /* the node count lives in m0: lane select of the draw and of the path word */ \
    "ds_read_b32 v75, %[lc] offset:" WA_LC_OFF_PARAM "\n"          /* lanes 5, 6 of the parameter column: kmul, 16 - hl (16-bit entries) */ \
    "s_mov_b32 s71, 0xffff\n"                                     /* the empty 16-bit entry */                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "v_readlane_b32 s68, v75, 5\n"                                                                                \
    "v_readlane_b32 s69, v75, 6\n"                                                                                \
    "s_nop 3\n"                                                                                                   \
    WA_ASM_NEXT_PROBE_T16                                                                                         \
    "s_mul_i32 s40, %[cur], 24\n"                /* records of cur's six neighbours: what the first step's tail would have requested */ \
    "v_add_u32 v82, s40, v65\n"                                                                                   \
    "global_load_dword v73, v82, %[pher]\n"                                                                       \
    "global_load_dword v74, v82, %[heur]\n"
#if defined(WA_ASM_STAMPS)
#if defined(WA_ASM_SPAN_A)
#define WA_ASM_STAMPS_INIT "s_mov_b32 s72, 0\n s_mov_b32 s73, 0\n s_mov_b32 s74, 0\n s_mov_b32 s75, 0\n s_mov_b32 s76, 0\n s_mov_b32 s77, 0\n" \
                           "s_mov_b32 s60, 0\n s_mov_b32 s62, 0\n"
#else
#define WA_ASM_STAMPS_INIT "s_mov_b32 s72, 0\n s_mov_b32 s73, 0\n s_mov_b32 s74, 0\n s_mov_b32 s75, 0\n s_mov_b32 s76, 0\n s_mov_b32 s77, 0\n" \
                           "s_memtime s[60:61]\n s_waitcnt lgkmcnt(0)\n s_mov_b32 s70, s60\n"
#endif
#define WA_ASM_STAMPS_DUMP "v_mov_b32 v94, s72\n ds_write_b32 %[lc], v94 offset:1792\n v_mov_b32 v94, s73\n ds_write_b32 %[lc], v94 offset:2048\n" \
                           "v_mov_b32 v94, s74\n ds_write_b32 %[lc], v94 offset:2304\n v_mov_b32 v94, s75\n ds_write_b32 %[lc], v94 offset:2560\n" \
                           "v_mov_b32 v94, s76\n ds_write_b32 %[lc], v94 offset:2816\n v_mov_b32 v94, s77\n ds_write_b32 %[lc], v94 offset:3072\n"
#define WA_ASM_STAMPS_CLOBBER "s60", "s61", "s62", "s63", "s64", "s70", "s72", "s73", "s74", "s75", "s76", "s77",
#else
#define WA_ASM_STAMPS_INIT ""
#define WA_ASM_STAMPS_DUMP ""
#define WA_ASM_STAMPS_CLOBBER
#endif
// The straggler check (dense loops, one search per solver; TAG "f": at a block boundary, "p": inside a block, see the event handler): an ant whose node count already exceeds that of %[cutn]
// arrivals of its generation cannot be among the depositing ranks nor become the best path any more; it leaves with code 5 and is
// finished beside the next generation's ants (see k_walk_dev).  %[cut] = node counts of the arrivals so far (0xffffffff = none yet),
// lane l looks at entries l, l+64, l+128, l+192; %[cutn] = 0x7fffffff switches the check off.  v86..v89 are the touch loads' registers
// (nobody reads those): the loads in flight are drained first.
#define WA_ASM_CUT(TAG)                                                                                           \
    "s_cmp_eq_u32 %[cutn], 0x7fffffff\n"                                                                          \
    "s_cbranch_scc1 Lwa_nocut_" TAG "%=\n"                                                                        \
    "s_waitcnt vmcnt(0)\n"                                                                                        \
    "v_lshlrev_b32 v94, 2, v64\n"                                                                                 \
    "global_load_dword v86, v94, %[cut] sc1\n"                                                                    \
    "global_load_dword v87, v94, %[cut] offset:256 sc1\n"                                                         \
    "global_load_dword v88, v94, %[cut] offset:512 sc1\n"                                                         \
    "global_load_dword v89, v94, %[cut] offset:768 sc1\n"                                                         \
    "s_mov_b32 s46, m0\n"                                                                                         \
    "s_waitcnt vmcnt(0)\n"                                                                                        \
    "v_cmp_gt_u32 vcc, s46, v86\n s_nop 4\n s_bcnt1_i32_b64 s58, vcc\n"                                          \
    "v_cmp_gt_u32 vcc, s46, v87\n s_nop 4\n s_bcnt1_i32_b64 s59, vcc\n s_add_u32 s58, s58, s59\n"                \
    "v_cmp_gt_u32 vcc, s46, v88\n s_nop 4\n s_bcnt1_i32_b64 s59, vcc\n s_add_u32 s58, s58, s59\n"                \
    "v_cmp_gt_u32 vcc, s46, v89\n s_nop 4\n s_bcnt1_i32_b64 s59, vcc\n s_add_u32 s58, s58, s59\n"                \
    "s_cmp_lg_u32 s58, 0\n"                                                                                       \
    "s_cselect_b32 %[em], " WA_CUT_FINE_MASK ", %[em]\n"            /* shorter ants have arrived: from now on look every 16 steps */ \
    "s_cmp_ge_u32 s58, %[cutn]\n"                                                                                 \
    "s_cbranch_scc0 Lwa_nocut_" TAG "%=\n"                                                                        \
    "s_mov_b32 %[code], 5\n"                                                                                      \
    "s_branch Lwa_out%=\n"                                                                                        \
    "Lwa_nocut_" TAG "%=:\n"
#define WA_ASM_NOCUT(TAG) ""
#ifndef WA_CUT_FINE_MASK
#define WA_CUT_FINE_MASK "15"
#endif
// everything behind the loop: the dead-end exit and the pending-event handler (a completed 64-word block and/or the arrival)
#define WA_ASM_TAIL WA_ASM_TAIL_(WA_ASM_NOCUT)
#define WA_ASM_TAIL_(CUT)                                                                                         \
    "Lwa_dead%=:\n"                                                                                               \
    "s_mov_b32 %[code], 1\n"                                                                                      \
    "s_branch Lwa_out%=\n"                                                                                        \
    "Lwa_event%=:\n"                                                                                              \
    WA_ASM_COUNT_EVENT                                                                                            \
    "s_mov_b32 %[code], 2\n"                                                                                      \
    "s_and_b32 s46, m0, 63\n"                                                                                     \
    "s_cbranch_scc0 Lwa_full%=\n"                                                                                 \
    "s_cmp_eq_u32 %[cur], %[end]\n"                                                                               \
    "s_cbranch_scc1 Lwa_out%=\n"                   /* no complete block, at the end point: the arrival (:182-186) */ \
    "s_mov_b32 %[code], 0\n"                      /* inside a block: a straggler check was due (%[em] = 15) */    \
    CUT("p")                                                                                                      \
    "s_branch Lwa_resume%=\n"                                                                                     \
    "Lwa_full%=:\n"                                                                                               \
    "s_lshl_b32 s46, m0, 2\n"                      /* block [len-64, len) -> path[]: one coalesced 256-byte store */ \
    "v_lshlrev_b32 v94, 2, v64\n"                                                                                 \
    "v_add_u32 v94, s46, v94\n"                                                                                   \
    "v_add_u32 v94, 0xffffff00, v94\n"                                                                            \
    "global_store_dword v94, %[pbuf], %[path]\n"                                                                  \
    "s_cmp_eq_u32 %[cur], %[end]\n"                                                                               \
    "s_cbranch_scc1 Lwa_out%=\n"                   /* ... and arrived with it (code 2) */                         \
    "s_mov_b32 %[code], 0\n"                                                                                      \
    "s_add_i32 s46, m0, 64\n"                                                                                     \
    "s_cmp_gt_i32 s46, %[limit]\n"                                                                                \
    "s_cbranch_scc1 Lwa_out%=\n"                   /* the next block would pass the table-load / capacity limit: the caller's generic loop goes on */ \
    CUT("f")                                                                                                      \
    "v_add_u32 v94, m0, v64\n"                     /* the next 64 draws: lane i <- draw of step len + i - 1 (wa_ctr_draw) */ \
    "v_add_u32 v94, -1, v94\n"                                                                                    \
    "s_mov_b32 s46, 0x9e3779b9\n"                                                                                 \
    "v_mul_lo_u32 v94, v94, s46\n"                                                                                \
    "v_add_u32 v94, %[klo], v94\n"                                                                                \
    "v_xor_b32 v94, %[khi], v94\n"                                                                                \
    "v_lshrrev_b32 v95, 16, v94\n"                                                                                \
    "v_xor_b32 v94, v95, v94\n"                                                                                   \
    "s_mov_b32 s46, 0x7feb352d\n"                                                                                 \
    "v_mul_lo_u32 v94, v94, s46\n"                                                                                \
    "v_lshrrev_b32 v95, 15, v94\n"                                                                                \
    "v_xor_b32 v94, v95, v94\n"                                                                                   \
    "s_mov_b32 s46, 0x846ca68b\n"                                                                                 \
    "v_mul_lo_u32 v94, v94, s46\n"                                                                                \
    "v_lshrrev_b32 v95, 16, v94\n"                                                                                \
    "v_xor_b32 v94, v95, v94\n"                                                                                   \
    "v_lshrrev_b32 v94, 1, v94\n"                                                                                 \
    "v_cvt_f32_u32 v94, v94\n"                                                                                    \
    "v_mul_f32 %[ub], 0x30000000, v94\n"           /* (float)r / 2^31 (:169) */                                   \
    "Lwa_resume%=:\n"                                                                                             \
    "s_lshl_b64 s[54:55], 63, %[g8]\n"             /* the active mask back: evaluate the pending step again */    \
    "s_cmp_eq_u32 s47, 0\n"                                                                                       \
    "s_cbranch_scc1 Lwa_redo_a%=\n"                                                                               \
    "s_cmp_eq_u32 s47, 1\n"                                                                                       \
    "s_cbranch_scc1 Lwa_redo_b%=\n"                                                                               \
    "s_cmp_eq_u32 s47, 2\n"                                                                                       \
    "s_cbranch_scc1 Lwa_redo_c%=\n"                                                                               \
    "s_branch Lwa_redo_d%=\n"                                                                                     \
    "Lwa_ovf%=:\n"                                                /* (16-bit entries) a probe chain ran past what an entry can say: the walk spills */ \
    "s_mov_b32 %[code], 6\n"                                                                                      \
    "Lwa_out%=:\n"                                                                                                \
    "s_mov_b32 %[len], m0\n"                                                                                      \
    WA_ASM_STAMPS_DUMP                                                                                            \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
#define WA_ASM_CLOBBERS                                                                                           \
    "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v100", "v101", "v76", "v77", "s68", "s69", "s71", "v78", "v79", "v80", "v81", "v82",  \
        "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s40", "s41", "s42", "s43", "s44",     \
        "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", WA_ASM_STAMPS_CLOBBER    \
        "vcc", "scc", "m0", "memory"

// Where the loop sits relative to the 64-byte instruction lines moves a step by up to 2 % (its branch targets: four step heads, the
// collision and rare-event stubs).  Measured at C3, walk launches of generations 0-10, loop top at 64 B + 4k bytes: k = 0 190.0 us,
// 2 188.7, 4 187.1, 6 187.5, 8 190.4, 10 189.7, 12 186.7, 14 187.1 -- so the dense loops are pinned at k = 12 instead of left to
// whatever the code in front of them happens to add up to.
// (round 3, after the loop changed -- probe first, ds_read2, collision re-entry label: k = 0..15 measured again with tools/walk_ab.py, 159.1-163.2 ns
// per step of the longest walk; 5, 12 and 13 are the best within noise, 12 stays)
#ifndef WA_ALIGN_PAD
#define WA_ALIGN_PAD 12
#endif
#define WA_ASM_NOPS_0 ""
#define WA_ASM_NOPS_1 " s_nop 0\n"
#define WA_ASM_NOPS_2 WA_ASM_NOPS_1 WA_ASM_NOPS_1
#define WA_ASM_NOPS_3 WA_ASM_NOPS_2 WA_ASM_NOPS_1
#define WA_ASM_NOPS_4 WA_ASM_NOPS_2 WA_ASM_NOPS_2
#define WA_ASM_NOPS_5 WA_ASM_NOPS_4 WA_ASM_NOPS_1
#define WA_ASM_NOPS_6 WA_ASM_NOPS_4 WA_ASM_NOPS_2
#define WA_ASM_NOPS_7 WA_ASM_NOPS_4 WA_ASM_NOPS_3
#define WA_ASM_NOPS_8 WA_ASM_NOPS_4 WA_ASM_NOPS_4
#define WA_ASM_NOPS_9 WA_ASM_NOPS_8 WA_ASM_NOPS_1
#define WA_ASM_NOPS_10 WA_ASM_NOPS_8 WA_ASM_NOPS_2
#define WA_ASM_NOPS_11 WA_ASM_NOPS_8 WA_ASM_NOPS_3
#define WA_ASM_NOPS_12 WA_ASM_NOPS_8 WA_ASM_NOPS_4
#define WA_ASM_NOPS_13 WA_ASM_NOPS_12 WA_ASM_NOPS_1
#define WA_ASM_NOPS_14 WA_ASM_NOPS_12 WA_ASM_NOPS_2
#define WA_ASM_NOPS_15 WA_ASM_NOPS_12 WA_ASM_NOPS_3
#define WA_ASM_CAT_(a, b) a##b
#define WA_ASM_CAT(a, b) WA_ASM_CAT_(a, b)
#define WA_ASM_LOOP_ALIGN ".p2align 6\n" WA_ASM_CAT(WA_ASM_NOPS_, WA_ALIGN_PAD)
#define WA_ASM_REJ_INIT "s_mov_b32 s78, -1\n v_readlane_b32 s80, v94, 3\n v_readlane_b32 s81, v94, 4\n"
#define WA_ASM_REJ_EXITS WA_ASM_REJ_EXIT("v71", "v72", "a") WA_ASM_REJ_EXIT("v73", "v74", "b") WA_ASM_REJ_EXIT("v71", "v72", "c") WA_ASM_REJ_EXIT("v73", "v74", "d")
// the two dense loops as statements (W = SELF | NONE: the touch loads, see above)
#define WA_ASM_RUN_DENSE(W, T)                                                                                       \
    asm volatile(                                                                                                 \
        WA_ASM_PROLOGUE_##T WA_ASM_STAMPS_INIT WA_ASM_LOOP_ALIGN                                                      \
        "Lwa_top%=:\n"                                                                                            \
        WA_ASM_STEP("v71", "v72", "v73", "v74", "a", W, T)                                                           \
        WA_ASM_STEP("v73", "v74", "v71", "v72", "b", W, T)                                                           \
        WA_ASM_STEP("v71", "v72", "v73", "v74", "c", W, T)                                                           \
        WA_ASM_STEP("v73", "v74", "v71", "v72", "d", W, T)                                                           \
        "s_branch Lwa_top%=\n"                                                                                    \
        WA_ASM_COLL("a", T) WA_ASM_COLL("b", T) WA_ASM_COLL("c", T) WA_ASM_COLL("d", T)                                       \
        WA_ASM_RARE("v71", "v72", "a", "0") WA_ASM_RARE("v73", "v74", "b", "1") WA_ASM_RARE("v71", "v72", "c", "2") WA_ASM_RARE("v73", "v74", "d", "3") \
        WA_ASM_TAIL_(WA_ASM_CUT)                                                                                  \
        : [code] "=&s"(code), [cur] "+s"(cur), [len] "+s"(len), [g8] "+s"(g8), [pio] "+v"(p), [hio] "+v"(h), [pbuf] "+v"(pbuf), [ub] "+v"(ublock), [em] "+s"(em) \
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [c6] "v"(c6), [lc] "v"(lcaddr), [pher] "s"(pher_b), [heur] "s"(heur_b), [hs] "s"(hshift), [hm4] "s"(hm4), [end] "s"(end), [path] "s"(path), \
          [limit] "s"(limit), [klo] "s"((uint32_t)antkey), [khi] "s"((uint32_t)(antkey >> 32)), [cut] "s"(cut_list), [cutn] "s"(cut_n) \
        : WA_ASM_CLOBBERS);
#define WA_ASM_RUN_REJ(W, T)                                                                                         \
    asm volatile(                                                                                                 \
        WA_ASM_PROLOGUE_##T                                                                                       \
        "ds_read_b32 v94, %[lc] offset:" WA_LC_OFF_PARAM "\n"        /* lanes 3, 4: best-path version, hold-off */               \
        "s_waitcnt lgkmcnt(0)\n"                                                                                  \
        WA_ASM_REJ_INIT WA_ASM_LOOP_ALIGN                                                                         \
        "Lwa_top%=:\n"                                                                                            \
        WA_ASM_STEP_REJ("v71", "v72", "v73", "v74", "a", W, T)                                                       \
        WA_ASM_STEP_REJ("v73", "v74", "v71", "v72", "b", W, T)                                                       \
        WA_ASM_STEP_REJ("v71", "v72", "v73", "v74", "c", W, T)                                                       \
        WA_ASM_STEP_REJ("v73", "v74", "v71", "v72", "d", W, T)                                                       \
        "s_branch Lwa_top%=\n"                                                                                    \
        WA_ASM_COLL("a", T) WA_ASM_COLL("b", T) WA_ASM_COLL("c", T) WA_ASM_COLL("d", T)                                       \
        WA_ASM_RARE("v71", "v72", "a", "0") WA_ASM_RARE("v73", "v74", "b", "1") WA_ASM_RARE("v71", "v72", "c", "2") WA_ASM_RARE("v73", "v74", "d", "3") \
        WA_ASM_REJ_EXITS                                                                                          \
        WA_ASM_TAIL_(WA_ASM_CUT)                                                                                  \
        : [code] "=&s"(code), [cur] "+s"(cur), [len] "+s"(len), [g8] "+s"(g8), [pio] "+v"(p), [hio] "+v"(h), [pbuf] "+v"(pbuf), [ub] "+v"(ublock), [em] "+s"(em) \
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [c6] "v"(c6), [lc] "v"(lcaddr), [pher] "s"(pher_b), [heur] "s"(heur_b), [hs] "s"(hshift), [hm4] "s"(hm4), [end] "s"(end), [path] "s"(path), \
          [limit] "s"(limit), [klo] "s"((uint32_t)antkey), [khi] "s"((uint32_t)(antkey >> 32)), [markb] "s"(mark), [cut] "s"(cut_list), [cutn] "s"(cut_n) \
        : "s78", "s80", "s81", "s82", "s83", WA_ASM_CLOBBERS);
// the lazy-field loop as a statement (STEP = WA_ASM_STEP_LAZY or WA_ASM_STEP_LAZY_REJ; REJINIT / REJEXITS = the rejoin watch's
// set-up and hand-back stubs, empty without it; `mark` is passed either way)
#define WA_ASM_RUN_LAZY(STEP, W, REJINIT, REJEXITS, T)                                                                 \
    asm volatile(                                                                                                 \
        WA_ASM_PROLOGUE_##T                                                                                       \
        "ds_read_b32 v99, %[lc] offset:" WA_LC_OFF_STAMP "\n"        /* stamp offset of this lane's neighbour */                 \
        "ds_read_b32 v94, %[lc] offset:" WA_LC_OFF_PARAM "\n"        /* lanes 0..2: clean value, evap_now + 1, rho; 3, 4: version, hold-off */ \
        "v_mov_b32 v96, %[sio]\n"                                                                                 \
        "s_waitcnt lgkmcnt(0)\n"                                                                                  \
        "v_readlane_b32 s33, v94, 0\n"                                                                            \
        "v_readlane_b32 s34, v94, 1\n"                                                                            \
        "v_readlane_b32 s35, v94, 2\n"                                                                            \
        REJINIT                                                                                                   \
        "s_lshl_b32 s46, %[cur], 2\n"                                                                             \
        "v_add_u32 v98, s46, v99\n"                                                                               \
        "global_load_dword v97, v98, %[stamp]\n"                                                                  \
        WA_ASM_STAMPS_INIT                                                                                        \
        "Lwa_top%=:\n"                                                                                            \
        STEP("v71", "v72", "v96", "v73", "v74", "v97", "a", W, T)                                                     \
        STEP("v73", "v74", "v97", "v71", "v72", "v96", "b", W, T)                                                     \
        STEP("v71", "v72", "v96", "v73", "v74", "v97", "c", W, T)                                                     \
        STEP("v73", "v74", "v97", "v71", "v72", "v96", "d", W, T)                                                     \
        "s_branch Lwa_top%=\n"                                                                                    \
        WA_ASM_COLL("a", T) WA_ASM_COLL("b", T) WA_ASM_COLL("c", T) WA_ASM_COLL("d", T)                                       \
        WA_ASM_DIRTY("v71", "v72", "a") WA_ASM_DIRTY("v73", "v74", "b") WA_ASM_DIRTY("v71", "v72", "c") WA_ASM_DIRTY("v73", "v74", "d") \
        "Lwa_rare_a%=:\n v_mov_b32 %[sio], v96\n" WA_ASM_RARE_BODY("v71", "v72", "0")                             \
        "Lwa_rare_b%=:\n v_mov_b32 %[sio], v97\n" WA_ASM_RARE_BODY("v73", "v74", "1")                             \
        "Lwa_rare_c%=:\n v_mov_b32 %[sio], v96\n" WA_ASM_RARE_BODY("v71", "v72", "2")                             \
        "Lwa_rare_d%=:\n v_mov_b32 %[sio], v97\n" WA_ASM_RARE_BODY("v73", "v74", "3")                             \
        REJEXITS                                                                                                  \
        WA_ASM_TAIL                                                                                               \
        : [code] "=&s"(code), [cur] "+s"(cur), [len] "+s"(len), [g8] "+s"(g8), [pio] "+v"(p), [hio] "+v"(h), [pbuf] "+v"(pbuf), [ub] "+v"(ublock), [em] "+s"(em), \
          [sio] "+v"(pd)                                                                                          \
        : [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [c6] "v"(c6), [lc] "v"(lcaddr), [pher] "s"(pher_b), [heur] "s"(heur_b), [hs] "s"(hshift), [hm4] "s"(hm4), [end] "s"(end), [path] "s"(path), \
          [limit] "s"(limit), [klo] "s"((uint32_t)antkey), [khi] "s"((uint32_t)(antkey >> 32)), [stamp] "s"(stamp_b), [markb] "s"(mark) \
        : "v96", "v97", "v98", "v99", "s33", "s34", "s35", "s36", "s37", "s78", "s80", "s81", "s82", "s83", WA_ASM_CLOBBERS);

// ---- REF mode on the hand-scheduled loop: the shared libc stream, 64 draws at a time.
// glibc TYPE_3 (random_r.c): r[f] += r[b], result r[f] >> 1, b = f - 3 (mod 31).  The walking wavefront keeps the state ROTATED so that the
// front word is lane 0 (lane j = r[(f + j) % 31]); the COUNT next outputs are then one fully unrolled pass with compile-time indices --
// 31 v_readlane into scalar registers, one s_add + one s_lshr + one v_writelane per output, 31 v_writelane back (rotated by COUNT so
// that the new front is lane 0 again): ~4 instructions per draw instead of the ~12 of one wa_glibc_next_lanes call per step, and none
// of them on the step's dependency chain.  Output n goes to lane LANE0 + n of `ub` as (float)r / 2^31 (ACSRank_3D.hpp:169).
// lane LANE (a constant) of v := a wave-uniform value (scalar data operand, immediate lane select: no wait states needed)
template <int LANE>
__device__ __forceinline__ int32_t wa_writelane_imm(int32_t v, int32_t val_uniform)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(val_uniform), "n"(LANE));
    return v;
}
template <int COUNT, int LANE0, int N = 0>
struct WaGlibcUnroll {
    static __device__ __forceinline__ void out(uint32_t (&r)[31], int32_t &raw)
    {
        constexpr int i = N % 31;
        r[i] += r[(i + 28) % 31];
        raw = wa_writelane_imm<LANE0 + N>(raw, (int32_t)(r[i] >> 1));
        WaGlibcUnroll<COUNT, LANE0, N + 1>::out(r, raw);
    }
    static __device__ __forceinline__ void back(const uint32_t (&r)[31], int32_t &rot)
    {
        if (N < 31) {
            rot = wa_writelane_imm<(N < 31 ? N : 0)>(rot, (int32_t)r[(N + COUNT) % 31]);
            WaGlibcUnroll<COUNT, LANE0, N + 1>::back(r, rot);
        }
    }
};
template <int COUNT, int LANE0>
struct WaGlibcUnroll<COUNT, LANE0, COUNT> {
    static __device__ __forceinline__ void out(uint32_t (&)[31], int32_t &) {}
    static __device__ __forceinline__ void back(const uint32_t (&r)[31], int32_t &rot)
    {
        if (COUNT < 31) {
            rot = wa_writelane_imm<(COUNT < 31 ? COUNT : 0)>(rot, (int32_t)r[(COUNT + COUNT) % 31]);
        }
    }
};
// (the outputs as integers, lane LANE0 + n = output n)
template <int COUNT, int LANE0>
__device__ __forceinline__ void wa_glibc_block_raw(int32_t &rot, int32_t &raw_out)
{
    uint32_t r[31];
#pragma unroll
    for (int j = 0; j < 31; j++) r[j] = (uint32_t)__builtin_amdgcn_readlane(rot, j);
    int32_t raw = 0;
    WaGlibcUnroll<COUNT, LANE0>::out(r, raw);
    WaGlibcUnroll<COUNT, LANE0>::back(r, rot);
    static_assert(COUNT >= 31, "the write-back recursion covers lanes 0..30 only when COUNT >= 31");
    raw_out = raw;
}
template <int COUNT, int LANE0>
__device__ __forceinline__ void wa_glibc_block(int32_t &rot, float &ub)
{
    uint32_t r[31];
#pragma unroll
    for (int j = 0; j < 31; j++) r[j] = (uint32_t)__builtin_amdgcn_readlane(rot, j);
    int32_t raw = 0;
    WaGlibcUnroll<COUNT, LANE0>::out(r, raw);
    WaGlibcUnroll<COUNT, LANE0>::back(r, rot);
    static_assert(COUNT >= 31, "the write-back recursion covers lanes 0..30 only when COUNT >= 31");
    ub = (float)raw / 2147483648.0f;
}
// canonical state (lane j = r[j], indices f, b) -> rotated / back
__device__ __forceinline__ int32_t wa_glibc_rotate(int32_t rs, int32_t f)
{
    const int lane = threadIdx.x;
    int src = f + lane;
    src = src >= 31 ? src - 31 : src;
    return lane < 31 ? __builtin_amdgcn_ds_bpermute(src * 4, rs) : 0;
}
__device__ __forceinline__ int32_t wa_glibc_unrotate(int32_t rot, int32_t f)
{
    const int lane = threadIdx.x;
    int src = lane - f;
    src = src < 0 ? src + 31 : src;
    return lane < 31 ? __builtin_amdgcn_ds_bpermute(src * 4, rot) : 0;
}

// take the canonical state (lane j = r[j], indices f, b) `k` outputs on, given the rotated states kept in front of every 64th output counted
// from where it stands (snap row i = 32 words in front of output 64 * i; read through L2: the caller may have written them a moment ago)
__device__ __forceinline__ void wa_glibc_seek(const int32_t *snap, int32_t k, int32_t &rs, int32_t &f, int32_t &b)
{
    const int lane = threadIdx.x;
    const int32_t rot = lane < 32 ? __hip_atomic_load(&snap[(k >> 6) * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    f = (f + (k & ~63)) % 31;
    rs = wa_glibc_unrotate(rot, f);
    b = f + 28;
    b = b >= 31 ? b - 31 : b;
    for (int32_t q = 0; q < (k & 63); q++) (void)wa_glibc_next_lanes(rs, f, b);
}

// VARIANT 0: dense field.  1 (LAZY): the field of a lazily evaporating solver (stamp per voxel, see WaAcsDev); `stamp` is the
// slot's stamp array, clean_info the value of a never-deposited admissible edge, evap_now the evaporations applied so far.
// 2: dense field + rejoin watch (3: lazy field + rejoin watch): `mark` / `ver` = best-path membership stamps, hold_off = steps before a rejoin is reported;
// st.reason = 4 when the loop handed back because the ant stood on the best path one step ago.
// REFDRAW (REF mode, dense field): the draws come from the libc stream (rng_rs / rng_f / rng_b, canonical form) instead of the counter hash.
// The loop only knows how to hash its next 64 draws, so it is given ONE block at a time: its limit is the end of the current block, it
// hands back at every block boundary (code 0, the block stored), the next 64 draws are generated here and it re-enters (its prologue
// re-requests the records: ~1 us per 64 steps).  A dead end is handed to the caller's generic loop undecided: the reference only calls
// rand() when a candidate exists (:162-166), and the loop's single exit does not say which of the two dead ends it met.
// DIRECT: no look-ahead (W = DIRECT above): every lane block requests the record of `cur` itself, the active block is always block 0
template <int VARIANT, bool WARM = true, bool REFDRAW = false, bool DIRECT = false, bool T16 = false>
__device__ __forceinline__ void wa_walk_fast_asm(const WaRun &R, const float *__restrict__ pher, const float *__restrict__ heur,
                                                 const uint32_t *__restrict__ stamp, float clean_info, uint32_t evap_now,
                                                 int32_t *path, int32_t *tab, int hash_log2, int32_t nx, int32_t nxy,
                                                 int32_t path_cap, int32_t end, uint64_t antkey, int32_t spill_at,
                                                 int32_t guard_bytes, int32_t stamp_guard_bytes, const float *__restrict__ ltab, WaWalkState &st,
                                                 int32_t *flags_out, const int32_t *prefix_words, unsigned long long *dbg,
                                                 const uint32_t *__restrict__ mark = nullptr, uint32_t ver = 0, int32_t hold_off = 0,
                                                 const uint32_t *cut_list = nullptr, int32_t cut_n = 0x7fffffff,
                                                 int32_t *rng_rs = nullptr, int32_t *rng_f = nullptr, int32_t *rng_b = nullptr,
                                                 uint32_t t16_kmul = 0)
{
    // T16: the table holds 16-bit entries (WaTabu in acs_walk.hpp; only the loops without touch loads know them); t16_kmul = their multiplier.
    // A compile-time choice: two inline statements merged under a run-time branch lose their scalar outputs to vector registers.
    static_assert(!T16 || (!WARM && !DIRECT && !REFDRAW), "16-bit tabu entries: the loops without touch loads only");
    t16_kmul = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)t16_kmul);
    constexpr bool t16 = T16;
    if (!cut_list) cut_n = 0x7fffffff;   // straggler check off (see WA_ASM_CUT); st.reason = 5 when the ant left through it
    cut_n = __builtin_amdgcn_readfirstlane(cut_n);   // (an SGPR operand of the loop: neither a literal nor a lane value)
    constexpr bool LAZY = VARIANT == 1 || VARIANT == 3, REJOIN = VARIANT == 2 || VARIANT == 3;
    const int lane = threadIdx.x;
    const int j = lane >> 3, pos = lane & 7;
    const int k2 = pos < 6 ? 5 - pos : 0;   // edge this lane evaluates; positions 6,7 of a group are padding (never admissible)
    // lane block b = lane >> 3 fetches the record of neighbour 5 - b: the block that becomes active after a move is
    // then the POSITION of the picked lane (edge k sits at position 5 - k), one s_lshl away from the pick
    const int32_t dk = wa_delta(k2, nx, nxy), dj = DIRECT ? 0 : wa_delta(j < 6 ? 5 - j : 5, nx, nxy);
    static_assert(!(DIRECT && WARM), "the touch loads belong to the look-ahead");
    const int32_t limit_full = path_cap < spill_at + 1 ? path_cap : spill_at + 1;
    int32_t limit = limit_full;   // (REFDRAW: the end of the current block, see below)
    const int32_t table = 1 << hash_log2;
    const int32_t tab_dw = t16 ? table / 2 : table;   // dwords the table takes: what lies behind it (sentinel, dummy slots, columns) starts there
    // field bases moved back by the guard band: every offset the loop forms is then non-negative
    const char *pher_b = reinterpret_cast<const char *>(pher) - guard_bytes;
    const char *heur_b = reinterpret_cast<const char *>(heur) - guard_bytes;
    const char *stamp_b = LAZY ? reinterpret_cast<const char *>(stamp) - stamp_guard_bytes : nullptr;
    // lane constants of the loop (operands; the loop copies them into its fixed registers v64..v70)
    const int32_t c0 = lane;
    const int32_t c1 = dj * 24 + k2 * 4 + guard_bytes;                          // record of neighbour j, edge k2
    const int32_t c2 = (dj + dk) * 24 + 4 + guard_bytes;                        // bytes 4..19 of the record two hops away
    const int32_t c3 = (int32_t)((uint32_t)dk * (t16 ? t16_kmul : 2654435761u));  // hash of (cur + dk) = cur*K + dk*K
    const int32_t c4 = dk;
    const int32_t c5 = (int32_t)((uint32_t)dk + ((uint32_t)k2 << WA_K_SHIFT));  // cur + this = path word of the move
    const int32_t c6 = (tab_dw + WA_WALK_LDS_PAD + lane) * 4;                   // this lane's dummy slot
    if (LAZY || REJOIN || t16) {   // two columns of 64 dwords behind the dummy slots
        int32_t *lc = tab + tab_dw + WA_WALK_LDS_PAD + 64;
        if (LAZY) lc[WA_LC_COL_STAMP * 64 + lane] = dj * 4 + stamp_guard_bytes;   // stamp of neighbour j
        // lanes 0..2: clean value, evap_now + 1, rho (lazy field); lanes 3, 4: best-path version, hold-off (rejoin watch)
        // ... lanes 5, 6: the multiplier and the second shift of the 16-bit entries
        lc[WA_LC_COL_PARAM * 64 + lane] = lane == 0 ? __float_as_int(clean_info) : lane == 1 ? (int32_t)(evap_now + 1u) : lane == 2 ? __float_as_int(R.rho)
                                          : lane == 3 ? (int32_t)ver : lane == 5 ? (int32_t)t16_kmul : lane == 6 ? 16 - hash_log2 : hold_off;
    }
    const int32_t lcaddr = (tab_dw + WA_WALK_LDS_PAD + 64 + lane) * 4;
    int32_t cur = __builtin_amdgcn_readfirstlane(st.cur), len = __builtin_amdgcn_readfirstlane(st.len), g8 = 0;   // (wave-uniform; the loop takes them as scalars)
    int32_t pbuf = st.cur;
    if (REJOIN && st.pbuf_valid) pbuf = st.pbuf;
    else if (prefix_words)   // (through L2: the words may have been stored by this very wavefront a moment ago)
        pbuf = lane < (st.len & 63) ? __hip_atomic_load(&prefix_words[(st.len & ~63) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    float ublock = 0.f;
    // REFDRAW: rot = the stream's state (rotated) behind the draws generated so far, rot_bs / len_bs = the same at the start of the
    // block the ant is in -- the canonical state handed back is that one advanced by the draws the ant really consumed
    int32_t rot = 0, rot_bs = 0, len_bs = len, f_bs = 0;
    if (REFDRAW) {
        f_bs = __builtin_amdgcn_readfirstlane(*rng_f);
        rot = rot_bs = wa_glibc_rotate(*rng_rs, f_bs);
        if ((len & 63) == 1) wa_glibc_block<63, 1>(rot, ublock);      // the walk starts at node count 1: steps 0..62 <-> lanes 1..63
        else {                                                      // (any other entry point: one draw per step, in order)
            int32_t rs = *rng_rs, f = f_bs, b = *rng_b, raw = 0;
            for (int q = len & 63; q < 64; q++) raw = wa_writelane(raw, wa_glibc_next_lanes(rs, f, b), q);
            rot = wa_glibc_rotate(rs, f);
            ublock = (float)raw / 2147483648.0f;
        }
    } else {
        ublock = (float)wa_ctr_draw(antkey, (uint32_t)((len & ~63) + lane - 1)) / 2147483648.0f;
    }
    float p = -0.f, h = 0.f;
    if (j == 0 && pos < 6) {
        const uint32_t boff = ((uint32_t)cur * 6u + (uint32_t)k2) * 4u;
        p = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pher) + boff);
        h = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(heur) + boff);
    }
    uint32_t pd = LAZY ? stamp[cur] : 1u;
    const int32_t hshift = t16 ? 31 - hash_log2 : 32 - hash_log2, hm4 = t16 ? (table - 1) * 2 : (table - 1) * 4;   // (16-bit entries: the slot's BYTE address = (t >> hshift) & hm4)
    int32_t em = 63;   // the loop raises an event whenever (node count & em) == 0: a completed block -- or, with 15, a straggler check inside one
    int exit_code;   // 1 dead end, 2 arrived, 3 leave the fast loop (table load / path capacity): the caller's generic loop goes on, 5 straggler
    for (;;) {
        // the loop checks its limits once per 64-step block (inside, when a block completes): only enter a block that fits entirely
        if ((len | 63) + 1 > limit_full) { exit_code = 3; break; }
        if (REFDRAW) limit = (len | 63) + 1;
        int32_t code;
        // (T16: 16-bit tabu entries -- saturated launches only, i.e. the loops without touch loads)
        if (REJOIN && !LAZY) {
            if (DIRECT) { WA_ASM_RUN_REJ(DIRECT, T32) } else if (WARM) { WA_ASM_RUN_REJ(SELF, T32) } else if constexpr (T16) { WA_ASM_RUN_REJ(NONE, T16) } else { WA_ASM_RUN_REJ(NONE, T32) }
        } else if (!LAZY) {
            if (DIRECT) { WA_ASM_RUN_DENSE(DIRECT, T32) } else if (WARM) { WA_ASM_RUN_DENSE(SELF, T32) } else if constexpr (T16) { WA_ASM_RUN_DENSE(NONE, T16) } else { WA_ASM_RUN_DENSE(NONE, T32) }
        } else if (REJOIN) {
            if (DIRECT) { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY_REJ, DIRECT, WA_ASM_REJ_INIT, WA_ASM_REJ_EXITS, T32) }
            else if (WARM) { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY_REJ, SELF, WA_ASM_REJ_INIT, WA_ASM_REJ_EXITS, T32) }
            else if constexpr (T16) { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY_REJ, NONE, WA_ASM_REJ_INIT, WA_ASM_REJ_EXITS, T16) }
            else { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY_REJ, NONE, WA_ASM_REJ_INIT, WA_ASM_REJ_EXITS, T32) }
        } else {
            if (DIRECT) { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY, DIRECT, "", "", T32) } else if (WARM) { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY, SELF, "", "", T32) }
            else if constexpr (T16) { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY, NONE, "", "", T16) } else { WA_ASM_RUN_LAZY(WA_ASM_STEP_LAZY, NONE, "", "", T32) }
        }
#if defined(WA_ASM_STAMPS)
        if (dbg && lane == 0) {
            const int32_t *lcs = tab + table + WA_WALK_LDS_PAD + 64;
            for (int i = 0; i < 6; i++) atomicAdd(&dbg[i], (unsigned long long)(uint32_t)lcs[(7 + i) * 64]);
        }
#endif
        if (code == 0) {           // stopped at a block boundary (the block is stored): the limit test above decides
            if (REFDRAW) {         // ... and the next block's 64 draws come from the libc stream
                f_bs += len - len_bs;
                f_bs %= 31;
                rot_bs = rot;
                len_bs = len;
                wa_glibc_block<64, 0>(rot, ublock);
            }
            continue;
        }
        exit_code = code;
        break;
    }
    if (REFDRAW) {
        // the stream continues behind the draws this ant consumed: one per step taken (a dead end met by the loop is re-evaluated by the
        // caller's generic loop, draw included if there is one)
        int32_t rs = wa_glibc_unrotate(rot_bs, f_bs), f = f_bs, b = f_bs + 28;
        b = b >= 31 ? b - 31 : b;
        for (int32_t q = len_bs; q < len; q++) (void)wa_glibc_next_lanes(rs, f, b);
        *rng_rs = rs; *rng_f = f; *rng_b = b;
        if (exit_code == 1) exit_code = 3;
    }
    st.reason = 0;
    if (exit_code == 4) {   // handed back by the rejoin watch, at the head of a step
        // a block completed by the last step is still in pbuf (its store belongs to the step that was not run); storing it
        // again after a boundary that was already handled is harmless.  (The watch looks once per step and never before the
        // first one -- s78 starts at -1 -- so at least one step was taken; the test on len keeps an empty pbuf from ever being
        // stored over a block that whoever brought the walk here had already written.)
        if ((len & 63) == 0 && len != st.len) path[(len - 64) + lane] = pbuf;
        if (cur == end) exit_code = 2;                       // ... which had arrived (:182-186)
        else if (len >= limit_full) exit_code = 3;
        else st.reason = 4;
    }
#if defined(WA_ASM_STAMPS)
    if (dbg && lane == 0) atomicAdd(&dbg[8], (unsigned long long)(len - st.len));
#endif
    if (exit_code == 5) st.reason = 5;
    if (exit_code == 6) {   // (16-bit entries) a probe chain ran past displacement 13: the loop cannot decide that lane
        // (left from inside a step: a block completed by the step before is still in pbuf -- its store belongs to the event this step had pending)
        if ((len & 63) == 0 && len != st.len) path[(len - 64) + lane] = pbuf;
        if (cur == end) exit_code = 2;   // ... and so is an arrival: the step that was being looked at is the one that would have found it (:182-186)
        else st.reason = 6;              // (the caller takes this one step with exact lookups -- fourteen slots decide -- and comes back)
    }
    st.done = exit_code != 3 && exit_code != 4 && exit_code != 5 && exit_code != 6;
    // :78, one add of `precision` per step taken (table).  A walk handed back by the rejoin watch does not need it yet: the
    // load would sit on the path of every re-entry
    float L = exit_code == 1 ? INFINITY : (exit_code == 4 || exit_code == 5) ? 0.f : ltab[len - 1];
    if (!st.done && len >= path_cap) {                     // the next step would not fit path[]
        if (lane == 0) atomicOr(flags_out, WA_FLAG_PATH_OVERFLOW);
        L = INFINITY;
        st.done = true;
    }
    if (len & 63) {  // partial last block (entries [len & ~63, len))
        if (lane < (len & 63)) path[(len & ~63) + lane] = pbuf;
    }
    st.cur = cur; st.len = len; st.step = (uint32_t)(len - 1); st.L = L;
    st.pbuf = (len & 63) ? pbuf : 0; st.pbuf_valid = true;
}


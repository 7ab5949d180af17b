// acs_kernels.hpp -- device side of ACS_Rank (ACSRank_3D.hpp), hand-written for gfx950 wave64.
//
//   k_init_pheromone  initFromGridMap :343-408 / reset :307-315         HBM write, 24 B/voxel
//   k_heuristic       the (1 + beta*cos) factor of selectNext :151-154  once per problem
//   k_begin           computeSolution prologue :229-233 + first :247-249
//   k_walk<DEV>       one WAVEFRONT per ant: lanes 0-5 own the six neighbours   latency-bound
//   k_walk<REF>       one wavefront per problem walks the ants in order on the libc stream
//   k_rank            best update :263-264, rank :273-275, deposit coefficients, next :247-249
//   k_evaporate       :268-272, float4 stream over 6N floats                     HBM-bound
//   k_deposit_mark/_apply   update_pheromone :198-215, rank-ordered adds without float atomics
//
// fp32 semantics are the reference's: IEEE div/sqrt, denormals kept, no FMA
// contraction (the TU is built with -ffp-contract=off), NaN propagation as written (Q3).
#pragma once
#include "wa_device.h"

#include "acs_dev.hpp"      // WaAcsDev, masks, straggler views; k_init_pheromone<NB>, k_heuristic<NB>, k_begin
#include "acs_walk.hpp"     // tabu, the walk loops, replay, k_walk_dev / k_walk_ref, k_replay_table / k_apply_table
#include "acs_update.hpp"   // k_rank, the sweep, k_evap_rank_mark, k_deposit_*, lazy helpers
#include "acs_nb26.hpp"     // 26-neighbour walk, table and apply kernels

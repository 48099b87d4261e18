// csrc/plan_math.h — host-only planning arithmetic (no HIP calls): transform
// length selection, the M = M1*M2 split, radix schedules, twiddle and index
// tables.  Kept separate so the CPU test-suite can exercise it without a GPU
// (tests/test_plan_math.py through the asx_planmath_* C entry points).
#pragma once

#include "asx_internal.h"

#include <string>
#include <vector>

struct AsxHostPlan {
    size_t N = 0;
    uint32_t F = 0, M = 0, src_valid = 0;
    int M1 = 0, M2 = 0, T = 0, logT = 0, ntiles = 0;
    AsxStages st1{}, st2{};
    std::vector<float2> tw1, tw2, tw2s, tw_lo, tw_hi;
    std::vector<int> k1_of_pos1, pos1_of_k1, pos2_of_k2;
    std::vector<int4> row_tasks;
    // real-column decomposition (rlayout.hip): possible when M1 is even, the length is not embedded and tiles are whole
    bool rlayout = false;
    std::vector<int4> col_pairs;    // [M1/2 + 1] {u, slot of u, slot of M1 - u, 0}, ordered by the slot of u
    std::vector<float2> col_tw;     // w_{2 M1}^u in the same order
};

// LDS budgets that bound the split (bytes per workgroup).
constexpr size_t ASX_LDS_COLS_MAX = 80 * 1024;  // M1 * T * 8  (two blocks per CU)
constexpr size_t ASX_LDS_ROWS_MAX = 64 * 1024;  // 4 * M2 * 8
constexpr size_t ASX_LDS_HW_MAX = 160 * 1024;   // what one gfx950 workgroup may declare (overrides only)

bool asx_is_smooth(uint64_t n);                 // only factors 2, 3, 5
uint64_t asx_next_smooth_even(uint64_t n);
// Fills *plan for sample_len N. `split_override` may be "" or "M1xM2xT".
// Returns "" on success, else an error message.
std::string asx_host_plan_build(size_t N, const char *split_override, AsxHostPlan *plan);
// The `max_count` cheapest splits of sample_len N by the planner's cost model, every feasible tile
// width included ("M1xM2xT" strings, cheapest first): the candidates of the measured mode
// (asx_plan_create_ex(..., "measure") times them on the device and keeps the fastest).
std::vector<std::string> asx_host_plan_candidates(size_t N, size_t max_count);
bool asx_make_stages(int n, AsxStages *st);
std::vector<int> asx_position_table(const AsxStages &st); // pos[k] = slot of X[k] after the DIF transform

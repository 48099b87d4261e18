// csrc/plan_math.cpp — see plan_math.h.  Pure host arithmetic.
#include <cstring>
#include <cstdlib>
#include <functional>
#include <algorithm>
#include "plan_math.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

bool asx_is_smooth(uint64_t n)
{
    if (n == 0) return false;
    for (uint64_t p : { 2ull, 3ull, 5ull })
        while (n % p == 0) n /= p;
    return n == 1;
}

uint64_t asx_next_smooth_even(uint64_t n)
{
    n += n & 1ull;
    if (n < 2) n = 2;
    while (!asx_is_smooth(n)) n += 2;
    return n;
}

// Radix schedule: the fewest passes over LDS using radices the kernels implement
// (lds_fft.h), ties broken by the smallest radix sum.  Order: even radices first
// (largest first), odd radices last -- the innermost stages (small q) then stride
// LDS by an odd number of elements, which spreads over the banks.
static const int kRadices[] = { 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2 };

static void search_radices(int rem, int depth, int sum, int *cur, int *best, int *best_depth, int *best_sum,
                           int min_next)
{
    if (rem == 1) {
        if (depth < *best_depth || (depth == *best_depth && sum < *best_sum)) {
            *best_depth = depth;
            *best_sum = sum;
            for (int i = 0; i < depth; i++) best[i] = cur[i];
        }
        return;
    }
    if (depth + 1 > *best_depth || depth >= ASX_MAX_STAGES) return;
    for (int R : kRadices) {
        if (R > min_next || rem % R) continue; // non-increasing: each multiset visited once
        cur[depth] = R;
        search_radices(rem / R, depth + 1, sum + R, cur, best, best_depth, best_sum, R);
    }
}

bool asx_make_stages(int n, AsxStages *st)
{
    *st = AsxStages{};
    st->n = n;
    int cur[ASX_MAX_STAGES], best[ASX_MAX_STAGES];
    int best_depth = ASX_MAX_STAGES + 1, best_sum = 1 << 30;
    if (n < 1) return false;
    search_radices(n, 0, 0, cur, best, &best_depth, &best_sum, 16);
    if (best_depth > ASX_MAX_STAGES) return false;
    // even radices (descending) first, then odd (descending)
    std::vector<int> order;
    for (int i = 0; i < best_depth; i++) if (best[i] % 2 == 0) order.push_back(best[i]);
    for (int i = 0; i < best_depth; i++) if (best[i] % 2 != 0) order.push_back(best[i]);
    // diagnostic: other orders of the same radices ([10,10,12] does fewer twiddle
    // multiplications than [12,10,10] but its innermost stage strides 12 slots = 4-way bank conflicts, +17 % k_rows)
    if (const char *e = getenv("ASX_STAGE_ORDER")) {
        if (!strcmp(e, "asc")) std::sort(order.begin(), order.end());
        else if (!strcmp(e, "desc")) std::sort(order.begin(), order.end(), std::greater<int>());
    }
    int ns = n;
    for (int i = 0; i < (int)order.size(); i++) {
        const int R = order[i];
        st->radix[i] = R;
        st->ns[i] = ns;
        st->q[i] = ns / R;
        st->nbf[i] = n / R;
        st->twmul[i] = n / ns;
        st->inv_q[i] = 1.0f / (float)(ns / R);
        st->inv_nbf[i] = 1.0f / (float)(n / R);
        ns /= R;
    }
    st->nstages = (int)order.size();
    return true;
}

std::vector<int> asx_position_table(const AsxStages &st)
{
    // k = u0 + R0*u1 + R0*R1*u2 + ...  ->  pos = u0*(n/R0) + u1*(n/(R0*R1)) + ...
    std::vector<int> pos(st.n);
    for (int k = 0; k < st.n; k++) {
        int rem = k, p = 0, stride = st.n;
        for (int i = 0; i < st.nstages; i++) {
            stride /= st.radix[i];
            p += (rem % st.radix[i]) * stride;
            rem /= st.radix[i];
        }
        pos[k] = p;
    }
    return pos;
}

static float2 unit_root(uint64_t num, uint64_t den)
{
    // exp(-2*pi*i*num/den) evaluated in double, rounded once to float
    num %= den;
    const double x = (double)num / (double)den;
    float2 w;
    if (num == 0) return make_float2(1.f, 0.f);
    if (4 * num == den) return make_float2(0.f, -1.f);
    if (2 * num == den) return make_float2(-1.f, 0.f);
    if (4 * num == 3 * den) return make_float2(0.f, 1.f);
    const double a = 2.0 * M_PI * x;
    w.x = (float)cos(a);
    w.y = (float)(-sin(a));
    return w;
}

static int pow2_floor(size_t v)
{
    int p = 1;
    while ((size_t)p * 2 <= v) p *= 2;
    return p;
}

static int tile_for(int M1)
{
    size_t cap = ASX_LDS_COLS_MAX / sizeof(float2) / (size_t)M1;
    if (cap < 1) return 0;
    int T = pow2_floor(cap);
    if (T > 64) T = 64;
    return T;
}

// transform length for sample_len N: 2N if that is {2,3,5}-smooth, else the next smooth even
// length >= 3N-1 (embedding, see asx_host_plan_build)
static uint64_t transform_len(size_t N)
{
    const uint64_t F = 2 * (uint64_t)N;
    return asx_is_smooth(F) ? F : asx_next_smooth_even(3 * (uint64_t)N - 1);
}

std::vector<std::string> asx_host_plan_candidates(size_t N, size_t max_count)
{
    std::vector<std::pair<double, std::string>> all;
    if (N == 0 || N > (size_t)1 << 27) return {};
    const uint64_t F = transform_len(N);
    if (F >= (1ull << 31)) return {};
    const uint32_t M = (uint32_t)(F / 2);
    const int rows_max = (int)(ASX_LDS_ROWS_MAX / (4 * sizeof(float2)));
    for (uint32_t a = 1; a <= M && a <= 8192u; a++) {
        if (M % a) continue;
        const uint32_t b = M / a;
        if ((int)b > rows_max) continue;
        AsxStages s1, s2;
        if (!asx_make_stages((int)a, &s1) || !asx_make_stages((int)b, &s2)) continue;
        int maxr = 2;
        for (int i = 0; i < s1.nstages; i++) maxr = std::max(maxr, s1.radix[i]);
        for (int i = 0; i < s2.nstages; i++) maxr = std::max(maxr, s2.radix[i]);
        const int tmax = tile_for((int)a);
        for (int t = 2; t <= tmax; t <<= 1) {
            if ((uint32_t)t > b && t > 2) continue;
            if (t < 8 && tmax >= 8) continue; // narrow tiles only where nothing wider fits
            const double lds = (double)std::max((size_t)a * t * sizeof(float2), (size_t)4 * b * sizeof(float2)) / 1024.0;
            const double cost = lds + 8.0 * (s1.nstages + s2.nstages) + (maxr > 12 ? 20.0 : 0.0);
            char buf[64];
            snprintf(buf, sizeof buf, "%ux%ux%d", a, b, t);
            all.emplace_back(cost, buf);
        }
    }
    std::sort(all.begin(), all.end());
    std::vector<std::string> out;
    for (size_t i = 0; i < all.size() && out.size() < max_count; i++) out.push_back(all[i].second);
    return out;
}

std::string asx_host_plan_build(size_t N, const char *split_override, AsxHostPlan *hp)
{
    if (N == 0) return "sample_len must be > 0";
    if (N > (size_t)1 << 27) return "sample_len too large";
    hp->N = N;
    uint64_t F = 2 * (uint64_t)N, valid = 2 * (uint64_t)N;
    if (!asx_is_smooth(F)) {
        // embed: r[k] = sum_n s'[n+k] sample[n] with s' the source extended periodically to
        // 3N-1 samples; any smooth even F >= 3N-1 gives the same r[0..2N) without wrap-around.
        valid = 3 * (uint64_t)N - 1;
        F = asx_next_smooth_even(valid);
    }
    if (F >= (1ull << 31)) return "transform length too large";
    hp->F = (uint32_t)F;
    hp->M = (uint32_t)(F / 2);
    hp->src_valid = (uint32_t)valid;
    const uint32_t M = hp->M;
    const int rows_max = (int)(ASX_LDS_ROWS_MAX / (4 * sizeof(float2)));

    int M1 = 0, M2 = 0, T = 0;
    // Measured-best splits for the reference's six interval lengths (src/audiosync.c:50-57) on
    // MI355X, from tools/tune.py (batched float32 path).  Other lengths use the cost model below.
    // The two longest lengths take 2400-point rows so that their column tiles are 400 / 600 packed rows of SIXTEEN real
    // columns (64-byte input pieces, whole 128-byte lines of the intermediates): with 800 / 1200 rows a tile holds eight,
    // and the real-column kernels (rlayout.hip) then re-fetch input lines between four sibling tiles (round 4, measured:
    // k_fwd_cols_r 6.15 GB against 5.09 GB per launch).  $ASX_LAYOUT=packed (the packed-sample kernels, xcorr_kernels.hip)
    // keeps round 3's table: its row kernels have no 2400-point schedule compiled in.
    static const struct { uint32_t M; const char *split; const char *split_packed; } kTuned[] = {
        { 144000u, "300x480x16", nullptr }, { 288000u, "600x480x16", nullptr }, { 480000u, "400x1200x16", nullptr },
        { 720000u, "600x1200x16", nullptr }, { 960000u, "400x2400x16", "800x1200x8" },
        { 1440000u, "600x2400x16", "1200x1200x8" },
    };
    if (!(split_override && *split_override)) {
        const char *lay = getenv("ASX_LAYOUT");
        const bool packed = (lay && !strcmp(lay, "packed")) || getenv("ASX_GENERIC");
        for (const auto &t : kTuned)
            if (t.M == M && F == 2 * (uint64_t)N) split_override = (packed && t.split_packed) ? t.split_packed : t.split;
    }
    if (split_override && *split_override) {
        if (sscanf(split_override, "%dx%dx%d", &M1, &M2, &T) != 3 || M1 < 1 || M2 < 1 ||
            (uint64_t)M1 * (uint64_t)M2 != M || T < 2 || (T & (T - 1)) || T > 64 ||
            (size_t)M1 * T * sizeof(float2) > ASX_LDS_HW_MAX ||
            (size_t)4 * M2 * sizeof(float2) > ASX_LDS_HW_MAX)
            return "bad ASX_SPLIT override";
    } else {
        // Cost of a split: the larger LDS footprint of the two kernel families (KiB; it bounds
        // the blocks per CU), 8 per pass over LDS (stage), 20 if any stage needs radix 15/16
        // (those bodies spill at 128 VGPRs).  Tiles of at least 8 columns (64-byte row
        // segments in HBM) are preferred; narrower ones only if nothing else fits.
        double best_cost = 1e30;
        for (int pass = 0; pass < 2 && M1 == 0; pass++) {
            const int min_T = pass == 0 ? 8 : 2;
            for (uint32_t a = 1; a <= M && a <= 8192u; a++) {
                if (M % a) continue;
                const uint32_t b = M / a;
                if ((int)b > rows_max) continue;
                int t = tile_for((int)a);
                if (t < min_T) continue;
                while (t > 2 && (uint32_t)t > b) t >>= 1; // not wider than the row (but always a column pair)
                AsxStages s1, s2;
                if (!asx_make_stages((int)a, &s1) || !asx_make_stages((int)b, &s2)) continue;
                int maxr = 2;
                for (int i = 0; i < s1.nstages; i++) maxr = std::max(maxr, s1.radix[i]);
                for (int i = 0; i < s2.nstages; i++) maxr = std::max(maxr, s2.radix[i]);
                const double lds = (double)std::max((size_t)a * t * sizeof(float2), (size_t)4 * b * sizeof(float2)) / 1024.0;
                const double cost = lds + 8.0 * (s1.nstages + s2.nstages) + (maxr > 12 ? 20.0 : 0.0);
                if (cost < best_cost || (cost == best_cost && t > T)) {
                    best_cost = cost;
                    M1 = (int)a; M2 = (int)b; T = t;
                }
            }
        }
        if (M1 == 0) return "transform length does not split into LDS-sized factors";
    }
    hp->M1 = M1; hp->M2 = M2; hp->T = T;
    hp->logT = 0;
    while ((1 << hp->logT) < T) hp->logT++;
    hp->ntiles = (M2 + T - 1) / T;
    if (!asx_make_stages(M1, &hp->st1) || !asx_make_stages(M2, &hp->st2))
        return "factor is not {2,3,5}-smooth";

    hp->tw1.resize(M1);
    for (int q = 0; q < M1; q++) hp->tw1[q] = unit_root(q, M1);
    hp->tw2.resize(M2);
    for (int q = 0; q < M2; q++) hp->tw2[q] = unit_root(q, M2);
    hp->tw_lo.resize(ASX_TW_LO);
    for (uint32_t q = 0; q < ASX_TW_LO; q++) hp->tw_lo[q] = unit_root(q, F);
    const uint32_t nhi = (uint32_t)((F + ASX_TW_LO - 1) >> ASX_TW_LOG) + 1;
    hp->tw_hi.resize(nhi);
    for (uint32_t h = 0; h < nhi; h++) hp->tw_hi[h] = unit_root((uint64_t)h << ASX_TW_LOG, F);

    hp->pos1_of_k1 = asx_position_table(hp->st1);
    hp->k1_of_pos1.assign(M1, 0);
    for (int k = 0; k < M1; k++) hp->k1_of_pos1[hp->pos1_of_k1[k]] = k;
    hp->pos2_of_k2 = asx_position_table(hp->st2);
    // w_M2^k2 in SLOT order (tw2s[pos2_of_k2[k2]] = w_M2^k2): the spectral combine of k_rows walks a row
    // slot by slot.  It relies on digit reversal mapping k2 -> M2-1-k2 to slot -> M2-1-slot (all digits
    // complemented); checked here so that a different position table cannot silently break it.
    hp->tw2s.resize(M2);
    for (int k2 = 0; k2 < M2; k2++) {
        if (hp->pos2_of_k2[M2 - 1 - k2] != M2 - 1 - hp->pos2_of_k2[k2]) return "position table is not a digit reversal";
        hp->tw2s[hp->pos2_of_k2[k2]] = hp->tw2[k2];
    }
    hp->row_tasks.resize(M1 / 2 + 1);
    for (int k1 = 0; k1 <= M1 / 2; k1++) {
        const int m1 = (M1 - k1) % M1;
        hp->row_tasks[k1] = make_int4(hp->pos1_of_k1[k1], hp->pos1_of_k1[m1], k1, m1);
    }
    // The real-column kernels (rlayout.hip) untangle the column transforms inside the tile: frequencies u and M1 - u,
    // u <= M1/2, pair up.  One entry per pair, {u, slot of u, slot of M1 - u}, ordered by the first slot so that
    // consecutive lanes walk LDS (almost) slot by slot; col_tw = w_{2 M1}^u in the same order.
    hp->col_pairs.clear();
    hp->col_tw.clear();
    hp->rlayout = false;
    if (M1 % 2 == 0 && F == 2 * (uint64_t)N && M2 % T == 0 && T >= 4 && M2 % 4 == 0) {
        std::vector<std::pair<int, int>> order; // (slot of u, u)
        for (int u = 0; u <= M1 / 2; u++) order.push_back({ hp->pos1_of_k1[u], u });
        std::sort(order.begin(), order.end());
        for (const auto &o : order) {
            const int u = o.second;
            hp->col_pairs.push_back(make_int4(u, o.first, hp->pos1_of_k1[(M1 - u) % M1], 0));
            hp->col_tw.push_back(unit_root((uint64_t)u, 2 * (uint64_t)M1));
        }
        hp->rlayout = true;
    }
    return "";
}

// csrc/shard_driver.cpp -- see shard_driver.h.  No HIP, no RCCL: plain C++ threads over an ops table.
#include "shard_driver.h"

#include <stdint.h>
#include <stdio.h>
#include <thread>

// 20 bytes per pair, rounded up to 8 so that every shard's record inside a gathered buffer starts 8-byte aligned (its
// int64 / double arrays can be read in place); the same rule as sharding.result_bytes.  (Found by this file's sanitizer
// test: round 3 returned 20 * width, i.e. misaligned records for odd widths.)
size_t asx_shard_record_bytes(size_t width) { return (width * (sizeof(int64_t) + sizeof(double) + sizeof(int32_t)) + 7) / 8 * 8; }

static std::string fmt(const char *f, int i, const std::string &why)
{
    char buf[96];
    snprintf(buf, sizeof buf, f, i);
    return std::string(buf) + why;
}

void asx_shard_release(const AsxShardOps &ops, AsxShardState &st)
{
    for (size_t i = 0; i < st.records.size(); i++)
        if (st.records[i]) { ops.free_record(ops.ctx, (int)i, st.records[i]); st.records[i] = nullptr; }
    st.width = 0;
}

int asx_shard_drive(const AsxShardOps &ops, AsxShardState &st, int n, const float *const *d_source, const float *const *d_sample,
                    const size_t *counts, size_t width, void *const *d_gathered, std::string *err)
{
    if (n < 1 || !d_source || !d_sample || !counts || !d_gathered || width == 0) { *err = "bad argument"; return -1; }
    for (int i = 0; i < n; i++)
        if (counts[i] > width || (counts[i] && (!d_source[i] || !d_sample[i])) || !d_gathered[i]) {
            *err = fmt("shard %d: count over width, or a null pointer", i, "");
            return -1;
        }
    const size_t rec = asx_shard_record_bytes(width);
    if (st.records.size() != (size_t)n) { asx_shard_release(ops, st); st.records.assign((size_t)n, nullptr); }
    if (width != st.width) {
        // the shards' own records, (re)sized to the width in use; a failure half-way leaves no record that a later call
        // would take for a sized one
        asx_shard_release(ops, st);
        for (int i = 0; i < n; i++) {
            std::string why;
            if (ops.alloc_record(ops.ctx, i, rec, &st.records[(size_t)i], &why) != 0) {
                asx_shard_release(ops, st);
                *err = fmt("shard %d: result record: ", i, why);
                return -1;
            }
        }
        st.width = width;
    }
    // phase 1: every device its shard, from its own host thread, on its stream (asynchronous)
    std::vector<int> rc((size_t)n, 0);
    std::vector<std::string> why((size_t)n);
    {
        std::vector<std::thread> workers;
        workers.reserve((size_t)n);
        for (int i = 0; i < n; i++)
            workers.emplace_back([&, i]() {
                rc[(size_t)i] = ops.run_shard(ops.ctx, i, st.records[(size_t)i], width, counts[i], d_source[i], d_sample[i],
                                              &why[(size_t)i]);
            });
        for (std::thread &t : workers) t.join(); // always all of them, whatever any one returned
    }
    int bad = -1;
    for (int i = 0; i < n && bad < 0; i++)
        if (rc[(size_t)i] != 0) bad = i;
    std::string gwhy;
    int grc = 0;
    // phase 2: ONE all-gather of the records, behind the kernels on the same streams
    if (bad < 0) grc = ops.gather(ops.ctx, n, st.records.data(), d_gathered, rec, &gwhy);
    // Every stream is waited for on EVERY path: a caller that frees its buffers after an error must not race kernels of
    // the shards that did start, and a successful call returns with the gathered records complete.
    int src = 0;
    std::string swhy;
    for (int i = 0; i < n; i++) {
        std::string w;
        if (ops.sync(ops.ctx, i, &w) != 0 && src == 0) { src = -1; swhy = fmt("shard %d: ", i, w); }
    }
    if (bad >= 0) { *err = fmt("shard %d: ", bad, why[(size_t)bad]); return -1; }
    if (grc != 0) { *err = "all-gather of the result records failed: " + gwhy; return -1; }
    if (src != 0) { *err = swhy; return -1; }
    return 0;
}

// csrc/shard_driver.h -- the control flow of asx_xcorr_batch_multi_dev (device-resident shards, one host
// thread and one stream per device, ONE all-gather of the result records), free of HIP and RCCL: every device operation
// goes through an ops table.  asx_api.hip binds the table to HIP + RCCL; tests/c/shard_driver_test.cpp binds it to
// host-memory stand-ins and runs this same code under ThreadSanitizer (no node with more than one GPU was available to
// rounds 1-4, so this is how the thread / partition / record logic is exercised).
#pragma once

#include <stddef.h>
#include <string>
#include <vector>

struct AsxShardOps {
    void *ctx;
    // the shard's own result record on its device: (re)allocation and release.  0 = ok.
    int (*alloc_record)(void *ctx, int shard, size_t bytes, void **out, std::string *err);
    void (*free_record)(void *ctx, int shard, void *record);
    // zero the record and enqueue the shard's `count` pairs on the shard's stream (asynchronous): lag / coef / ret
    // are views of `record` (int64 lag[width] | double coef[width] | int32 ret[width]).  Called from the shard's own thread.
    int (*run_shard)(void *ctx, int shard, void *record, size_t width, size_t count, const float *d_source,
                     const float *d_sample, std::string *err);
    // one grouped all-gather of the records, enqueued behind the kernels on the shards' streams (one caller thread)
    int (*gather)(void *ctx, int nshards, void *const *records, void *const *gathered, size_t record_bytes, std::string *err);
    // wait for everything enqueued on the shard's stream.  0 = ok.
    int (*sync)(void *ctx, int shard, std::string *err);
};

struct AsxShardState {            // what persists between calls (owned by asx_comm)
    std::vector<void *> records;  // per shard, sized for `width` pairs; null = none
    size_t width = 0;             // 0 = the records are not usable as they are
};

size_t asx_shard_record_bytes(size_t width);

// Runs one sharded batch.  Returns 0, or -1 with *err set.  Whatever happens, when it returns no shard has work in
// flight that reads the caller's inputs or writes the records (every stream that was given work has been waited for).
int asx_shard_drive(const AsxShardOps &ops, AsxShardState &st, int nshards, const float *const *d_source,
                    const float *const *d_sample, const size_t *counts, size_t width, void *const *d_gathered,
                    std::string *err);
void asx_shard_release(const AsxShardOps &ops, AsxShardState &st);

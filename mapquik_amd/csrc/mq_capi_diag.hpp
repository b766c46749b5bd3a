// mq_capi_diag.hpp -- measurement and diagnostic entry points (include/mapquik_hip_diag.h): probe statistics, stage clocks, launch timers, the
// random-probe rate of the memory system (part of the one translation unit mq_capi.hip).  Nothing here is on the product path.
#pragma once

// Diagnostic (tools/probe_rate.py): how many random index probes per second the memory system sustains, detached from
// everything else the map path does.  Every thread looks up `per_thread` pseudo-random keys (absent with probability ~1,
// like ~85 % of a read's k-min-mers), `ilp` home-slot loads in flight per thread.
__global__ void probe_rate_kernel(const Bucket *__restrict__ table, uint64_t mask, uint32_t per_thread, uint64_t seed,
                                  unsigned long long *__restrict__ acc, const uint32_t *__restrict__ bitmap, uint64_t bit_mask,
                                  uint32_t table_too) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nb = (mask + 1) >> 1;
    unsigned long long found = 0, steps = 0;
    auto mix = [](uint64_t z) {
        z += 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    };
    for (uint32_t j = 0; j < per_thread; j += 4) {
        uint64_t key[4];
        uint4 kk[4];
        uint32_t bw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            key[u] = mix(seed + tid * per_thread + j + u) | 1ull;
            if (bitmap) bw[u] = bitmap[(key[u] & bit_mask) >> 5];
            else kk[u] = ld_u4(&table[(key[u] & mask) >> 1].key[0]);
        }
        if (bitmap) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool maybe = (bw[u] >> (key[u] & 31u)) & 1u;
                found += maybe;
                kk[u] = (maybe && table_too) ? ld_u4(&table[(key[u] & mask) >> 1].key[0]) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // the probe sequence of mq_device.hpp: home way, other way, then the following buckets
            const uint32_t w0 = (uint32_t)(key[u] & mask) & 1u;
            uint64_t ka = u64_of(kk[u].x, kk[u].y), kb = u64_of(kk[u].z, kk[u].w);
            uint64_t kh = w0 ? kb : ka, kp = w0 ? ka : kb;
            uint64_t b = (key[u] & mask) >> 1;
            bool hit = kh == key[u], go = !hit && kh != 0;
            if (go) {
                steps++;
                hit = kp == key[u];
                go = !hit && kp != 0;
            }
            while (go) {
                b = b + 1 == nb ? 0 : b + 1;
                const uint4 v = ld_u4(&table[b].key[0]);
                ka = u64_of(v.x, v.y);
                kb = u64_of(v.z, v.w);
                steps++;
                hit = ka == key[u];
                go = !hit && ka != 0;
                if (go) {
                    steps++;
                    hit = kb == key[u];
                    go = !hit && kb != 0;
                }
            }
            if (hit) found += table[b].pay[0].start & 1u;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        found += __shfl_xor(found, d, 64);
        steps += __shfl_xor(steps, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], found);
        atomicAdd(&acc[1], steps);
    }
}

extern "C" {

int mq_last_map_path_counts(mq_index *idx, uint32_t *n_fast, uint32_t *n_general) try {
    if (!idx || !n_fast || !n_general) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    uint32_t v[2] = {0, 0};
    HIPCHK(hipMemcpy(v, c->d_counter + 4, 8, hipMemcpyDeviceToHost));
    *n_fast = v[0];
    *n_general = v[1];
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_last_map_order(mq_index *idx, uint32_t *n_flagged, uint32_t *n_first) try {
    if (!idx || !n_flagged || !n_first) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    uint32_t v = 0;
    HIPCHK(hipMemcpy(&v, c->d_counter + WORK_NF, 4, hipMemcpyDeviceToHost));
    *n_flagged = v;
    *n_first = v < WORK_FRONT_CAP ? v : WORK_FRONT_CAP;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_table_alloc_ms(mq_index *idx, float *ms) try {
    if (!idx || !ms) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    *ms = (float)idx->table_alloc_ms;
    return MQ_OK;
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_launch_waves(mq_index *idx, uint32_t n_reads, uint32_t *n_waves) try {
    if (!idx || !n_waves) return set_err(MQ_EINVAL, "bad arguments");
    int rc = use_device(idx);
    if (rc) return rc;
    if ((rc = ensure_geometry(idx))) return rc;
    const uint32_t grid = std::min<uint32_t>(idx->grid_fused, (n_reads + MAP_WAVES - 1) / MAP_WAVES);  // launch_map's grid
    *n_waves = grid * (uint32_t)MAP_WAVES;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_probe_stats(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases, mq_hit *d_out,
                       uint64_t *lookups, uint64_t *extra_steps) try {
    if (!idx || !lookups || !extra_steps) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    int rc = ctx_map_device(c, d_bases, d_offsets, n, total_bases, d_out, nullptr, true);  // the choice travels with this launch: contexts never see it
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    uint64_t v[2];
    HIPCHK(hipMemcpy(v, c->d_counter + 8, 16, hipMemcpyDeviceToHost));
    *extra_steps = v[0];
    *lookups = v[1];
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// Diagnostic: what every read of the last mq_map_probe_stats launch cost its wave (shader-clock cycles from the read's first instruction to
// its result's store) and when the wave took it up (the 100-MHz constant clock): where a launch's tail comes from.
int mq_last_read_cycles(mq_index *idx, uint32_t n, uint32_t *cycles, uint64_t *start_ticks) try {
    if (!idx || !cycles || !start_ticks) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid || n > c->reads_cap) return set_err(MQ_ESTATE, "no instrumented launch of that size recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipMemcpy(cycles, c->mz_count, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(start_ticks, c->mz_base, (size_t)n * 8, hipMemcpyDeviceToHost));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// Diagnostic (-DMQ_STAGE_CLOCKS builds; zeros otherwise): shader-clock cycles the waves of the last map_kernel launch of the default
// context spent per stage, summed over waves (stage list: mq_device.hpp, mq_clk).
int mq_last_stage_clocks(mq_index *idx, uint64_t *out16) try {
    if (!idx || !out16) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipMemcpy(out16, c->d_counter + 16, MQ_N_CLK * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_last_map_ms(mq_index *idx, float *ms) try {
    if (!idx || !ms) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    mq_ctx *c = idx->def_ctx;
    if (!c->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    HIPCHK(hipEventSynchronize(c->ev1));
    HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_probe_rate(mq_index *idx, uint32_t blocks, uint32_t per_thread, uint32_t bitmap_log2, uint32_t table_too, float *ms,
                  uint64_t *lookups, uint64_t *extra_steps) try {
    if (!idx || !ms || !lookups || !extra_steps) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    unsigned long long *d_acc = nullptr;
    uint32_t *bm = nullptr;
    uint64_t bit_mask = 0;
    HIPCHK(hipMalloc((void **)&d_acc, 16));
    HIPCHK(hipMemset(d_acc, 0, 16));
    if (bitmap_log2) {  // a stand-in bitmap with one bit in eight set
        bit_mask = (1ull << bitmap_log2) - 1;
        HIPCHK(hipMalloc((void **)&bm, (size_t)1 << (bitmap_log2 - 3)));
        HIPCHK(hipMemset(bm, 0x10, (size_t)1 << (bitmap_log2 - 3)));
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    per_thread = (per_thread + 3u) & ~3u;
    hipLaunchKernelGGL(probe_rate_kernel, dim3(blocks), dim3(256), 0, 0, idx->table, idx->nslots - 1, per_thread, 1ull, d_acc, bm, bit_mask, table_too);  // warm-up
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe_rate_kernel, dim3(blocks), dim3(256), 0, 0, idx->table, idx->nslots - 1, per_thread, 0x1234567ull, d_acc, bm, bit_mask, table_too);
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    unsigned long long acc[2];
    HIPCHK(hipMemcpy(acc, d_acc, 16, hipMemcpyDeviceToHost));
    hipFree(d_acc);
    hipFree(bm);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *lookups = (uint64_t)blocks * 256ull * per_thread;
    *extra_steps = acc[1] / 2;  // two launches accumulated
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_last_map_ms(mq_ctx *ctx, float *ms) try {
    if (!ctx || !ms) return set_err(MQ_EINVAL, "bad arguments");
    if (!ctx->ev_valid) return set_err(MQ_ESTATE, "no map launch recorded");
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

}  // extern "C"

// mq_capi_index.hpp -- C ABI, index side: mq_index_new .. mq_index_finalize (Index::new, ref_extract + add_with_mer, get_count +
// into_read_only; src/index.rs:78-116, src/mers.rs:15-38) (part of the one translation unit mq_capi.hip).
#pragma once

extern "C" {

mq_index *mq_index_new(const mq_params *params, int device) try {
    if (!params) {
        set_err(MQ_EINVAL, "params is NULL");
        return nullptr;
    }
    if (params->l < 1 || params->l > MAX_L || params->k < 1 || params->k > MAX_K) {
        set_err(MQ_EINVAL, "unsupported k/l: need 1 <= l <= 64 and 1 <= k <= 32");
        return nullptr;
    }
    if (params->flags & ~(MQ_FLAG_FOLD_CASE | MQ_FLAG_FAST_KH | MQ_FLAG_SEED_VARIANT_MASK)) {
        set_err(MQ_EINVAL, "undefined bits in mq_params.flags");
        return nullptr;
    }
    const uint32_t variant = (params->flags & MQ_FLAG_SEED_VARIANT_MASK) >> MQ_FLAG_SEED_VARIANT_SHIFT;
    if ((variant & MQ_SEEDVAR_POS_RUN_END) && params->l < 2) {
        set_err(MQ_EINVAL, "seeding variant 8 (position = end of the homopolymer run) needs l >= 2: the run's end is read off the window's second base");
        return nullptr;
    }
    const bool init_timing = getenv("MQ_DRIVER_TIMING") != nullptr;  // diagnostic (stderr): where the first index's start-up time goes
    const auto ti0 = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (init_timing) fprintf(stderr, "    mq_index_new: +%.3f s %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - ti0).count(), what);
    };
    int n = mq_device_count();
    stamp("hipGetDeviceCount (runtime initialised)");
    if (n <= 0) {
        set_err(MQ_ENODEVICE, "no HIP device: the mapquik HIP path has no CPU fallback");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_err(MQ_EINVAL, "device ordinal out of range");
        return nullptr;
    }
    mq_index *idx = new mq_index();
    idx->params = *params;
    idx->device = device;
    set_dev_bound(idx->dp, params->density, variant);
    idx->dp.k = params->k;
    idx->dp.l = params->l;
    idx->dp.use_hpc = params->use_hpc ? 1 : 0;
    idx->dp.c = params->c;
    idx->dp.s = params->s;
    idx->dp.g = params->g;
    idx->dp.fold = (params->flags & MQ_FLAG_FOLD_CASE) ? 1u : 0u;
    idx->dp.fast_kh = (params->flags & MQ_FLAG_FAST_KH) ? 1u : 0u;
    const char *cc = getenv("MQ_CHAIN_CHUNK");
    if (cc && atoi(cc) == 4) idx->chain_chunk = 4;
    const char *fg = getenv("MQ_FORCE_GENERAL");
    idx->force_general = fg && atoi(fg) != 0;
    const char *hf = getenv("MQ_HEAVY_FIRST");
    idx->heavy_first = !(hf && atoi(hf) == 0);
    const char *pl = getenv("MQ_PIPELINE");
    idx->split = pl && strcmp(pl, "split") == 0;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        set_err(MQ_EHIP, "hipSetDevice/hipGetDeviceProperties failed");
        delete idx;
        return nullptr;
    }
    idx->n_cu = prop.multiProcessorCount;
    stamp("hipSetDevice + hipGetDeviceProperties");
    // an empty one-bucket table so that seeding-only calls work before finalize
    if (alloc_table(idx, 2) != MQ_OK) {
        delete idx;
        return nullptr;
    }
    stamp("first hipMalloc + hipMemset (code objects loaded)");
    idx->def_ctx = ctx_create(idx);
    stamp("stream created");
    if (!idx->def_ctx) {
        hipFree(idx->table);
        delete idx;
        return nullptr;
    }
    return idx;
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

void mq_index_free(mq_index *idx) {
    if (!idx) return;
    rsv_join(idx);
    hipSetDevice(idx->device);
    if (idx->rsv_table) hipFree(idx->rsv_table);
    for (auto &c : idx->chunks)
        if (c.d) hipFree(c.d);
    free_build_scratch(idx);
    if (idx->table) hipFree(idx->table);
    if (idx->d_ref_lens) hipFree(idx->d_ref_lens);
    ctx_release(idx->def_ctx);
    delete idx;
}

static int64_t add_ref_device_locked(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *d_seq, uint64_t len) {
    if (!idx || (!d_seq && len)) return set_err(MQ_EINVAL, "bad arguments");
    if (idx->finalized) return set_err(MQ_ESTATE, "index already finalized");
    if (len >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
    if (ref_id >= MQ_MAX_REF_ID) return set_err(MQ_EINVAL, "ref_id must be < 2^24 (reference lengths are kept in a dense device array)");
    if (idx->refs.count(ref_id)) return set_err(MQ_EINVAL, "duplicate ref_id");
    int rc = use_device(idx);
    if (rc) return rc;
    idx->refs[ref_id] = std::make_pair(std::string(name ? name : ""), len);
    const DevParams &P = idx->dp;
    if (len < (uint64_t)P.l + P.k - 1) return 0;  // src/mers.rs:18
    if (P.keep_none) return 0;                    // (seeding variant 1 with a bound of 0: no l-mer passes `hash < 0`)
    const bool with_last = (P.variant & MQ_SEEDVAR_END_COMPRESSED) != 0;

    const uint32_t n_seg = (uint32_t)((len + REF_SEG - 1) / REF_SEG);
    // expected minimizers per segment: 2 * density of the compressed l-mers; cap with slack, worst case on retry
    double dens = idx->params.density;
    if (!(dens > 0)) dens = 0;
    if (dens > 1) dens = 1;
    uint32_t cap = (uint32_t)std::min<double>((double)REF_SEG, 3.0 * 2.0 * dens * (double)REF_SEG + 256.0);
    if (const char *e = getenv("MQ_REF_CAP")) cap = (uint32_t)std::max(1, atoi(e));  // test hook: tiny regions, so that segments take the redo path
    if ((rc = grow(idx->bld_counts, idx->bld_counts_cap, n_seg))) return rc;
    if ((rc = grow(idx->bld_queue, idx->bld_queue_cap, n_seg))) return rc;
    if ((rc = grow(idx->bld_seg_off, idx->bld_seg_off_cap, (uint64_t)n_seg + 1))) return rc;
    if (!idx->bld_info) HIPCHK(hipMalloc((void **)&idx->bld_info, 64));
    if (!idx->grid_ref) {
        int occ = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)seed_ref_kernel, 64 * SEED_WAVES, 0));
        idx->grid_ref = std::max(1, occ) * idx->n_cu;
    }
    unsigned long long info[2] = {0, 0};
    if ((rc = grow(idx->bld_seg_hash, idx->bld_seg_hash_cap, (uint64_t)n_seg * cap))) return rc;
    if ((rc = grow(idx->bld_seg_pos, idx->bld_seg_pos_cap, (uint64_t)n_seg * cap))) return rc;
    if (with_last && (rc = grow(idx->bld_seg_last, idx->bld_seg_last_cap, (uint64_t)n_seg * cap))) return rc;
    HIPCHK(hipMemsetAsync(idx->bld_info, 0, 64, 0));
    RefSeedArgs A;
    A.seq = d_seq;
    A.len = len;
    A.n_seg = n_seg;
    A.P = P;
    A.seg_hash = idx->bld_seg_hash;
    A.seg_pos = idx->bld_seg_pos;
    A.seg_last = with_last ? idx->bld_seg_last : nullptr;
    A.cap = cap;
    A.counts = idx->bld_counts;
    A.queue = idx->bld_queue;
    A.counters = reinterpret_cast<uint32_t *>(idx->bld_info + 2);
    A.force_general = idx->force_general ? 1u : 0u;
    {
        const uint32_t g1 = std::min<uint32_t>((uint32_t)idx->grid_ref, (n_seg + SEED_WAVES - 1) / SEED_WAVES);
        hipLaunchKernelGGL(seed_ref_kernel, dim3(g1), dim3(64 * SEED_WAVES), 0, 0, A);
        HIPCHK(hipGetLastError());
        // the queue's length lives on the device: a fixed grid, waves that find the queue empty leave at once
        hipLaunchKernelGGL(seed_ref_general_kernel, dim3(std::min<uint32_t>((uint32_t)idx->n_cu * 32u, n_seg)), dim3(64), 0, 0, A);
        HIPCHK(hipGetLastError());
        // (the fast seeder's queue is consumed by now: the same array takes the numbers of the segments whose list outgrew its region)
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, 0, idx->bld_counts, n_seg, cap, idx->bld_seg_off, idx->bld_info, idx->bld_queue);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy(info, idx->bld_info, 16, hipMemcpyDeviceToHost));
    }
    const uint32_t n_over = (uint32_t)info[1];
    const uint64_t n_mz = info[0];
    int64_t n_kmm = 0;
    if (n_mz >= P.k) {
        n_kmm = (int64_t)(n_mz - P.k + 1);
        if ((rc = grow(idx->bld_dense_hash, idx->bld_dense_hash_cap, n_mz))) return rc;
        if ((rc = grow(idx->bld_dense_pos, idx->bld_dense_pos_cap, n_mz))) return rc;
        if (with_last && (rc = grow(idx->bld_dense_last, idx->bld_dense_last_cap, n_mz))) return rc;
        uint32_t *const dense_last = with_last ? idx->bld_dense_last : nullptr;
        hipLaunchKernelGGL(compact_lists_kernel, dim3(std::min<uint32_t>(n_seg, 65535u)), dim3(64), 0, 0, idx->bld_seg_hash, idx->bld_seg_pos, cap,
                           idx->bld_counts, idx->bld_seg_off, n_seg, idx->bld_dense_hash, idx->bld_dense_pos, A.seg_last, dense_last);
        HIPCHK(hipGetLastError());
        if (n_over) {
            hipLaunchKernelGGL(seed_ref_redo_kernel, dim3(std::min<uint32_t>(n_over, (uint32_t)idx->n_cu * 32u)), dim3(64), 0, 0, A, idx->bld_queue, n_over,
                               idx->bld_seg_off, idx->bld_dense_hash, idx->bld_dense_pos, dense_last);
            HIPCHK(hipGetLastError());
        }
        // the reference's k-min-mers go behind those of the previous references in the current chunk while it has room
        if (idx->chunks.empty() || idx->chunks.back().n + (uint64_t)n_kmm > idx->chunks.back().cap) {
            KmmChunk ch;
            ch.cap = std::max<uint64_t>((uint64_t)n_kmm, 16ull << 20);
            HIPCHK(hipMalloc((void **)&ch.d, (size_t)ch.cap * sizeof(RefKmm)));
            idx->chunks.push_back(ch);
        }
        KmmChunk &ch = idx->chunks.back();
        const uint32_t kb = (uint32_t)std::min<uint64_t>(((uint64_t)n_kmm + 255) / 256, 65535ull);
        hipLaunchKernelGGL(ref_kminmers_kernel, dim3(kb), dim3(256), 0, 0, idx->bld_dense_hash, idx->bld_dense_pos, n_mz, P, ref_id, ch.d + ch.n, dense_last);
        HIPCHK(hipGetLastError());
        ch.n += (uint64_t)n_kmm;
        idx->n_kmm_total += (uint64_t)n_kmm;
    }
    return n_kmm;  // everything above runs on the null stream: the next call's kernels (and finalize) are ordered behind it
}

int64_t mq_index_add_ref_device(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *d_seq, uint64_t len) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (!getenv("MQ_BUILD_TIMING")) return add_ref_device_locked(idx, ref_id, name, d_seq, len);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t r = add_ref_device_locked(idx, ref_id, name, d_seq, len);
    hipDeviceSynchronize();
    idx->t_add_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return r;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int64_t mq_index_add_ref(mq_index *idx, uint32_t ref_id, const char *name, const uint8_t *seq, uint64_t len) try {
    if (!idx || (!seq && len)) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    if (len >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
    if ((rc = grow(idx->bld_seq, idx->bld_seq_cap, len + 64))) return rc;
    if (len) HIPCHK(hipMemcpy(idx->bld_seq, seq, len, hipMemcpyHostToDevice));
    return add_ref_device_locked(idx, ref_id, name, idx->bld_seq, len);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// ---- the reference file straight from file pieces (the batch form of src/closures.rs:46-94's reader loop for a caller that never holds
// a whole record in host memory): pieces of the file go to a device buffer of the file's size asynchronously, on a stream of their own,
// and a record is indexed from there (add_ref_device_locked on the null stream, made to wait for the pieces issued so far) while the
// pieces behind it are still on their way.  A human reference is 3.1 GB in 25 records: copied record by record from pageable memory
// it took 0.24 s of the driver's 0.31-s reference phase; streamed from page-locked 16-MB pieces it is the PCIe link's 0.07 s, hidden
// behind the file read.
int mq_index_stage_begin(mq_index *idx, uint64_t total_bytes) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    if (idx->finalized) return set_err(MQ_ESTATE, "index already finalized");
    int rc = use_device(idx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(idx->stg_mu);
    if (idx->stg_buf) return set_err(MQ_ESTATE, "mq_index_stage_begin: one staging buffer per index");
    HIPCHK(hipMalloc((void **)&idx->stg_buf, total_bytes + 64));
    const hipError_t es = hipStreamCreateWithFlags(&idx->stg_stream, hipStreamNonBlocking);
    if (es != hipSuccess) {  // no buffer without its stream: a later mq_index_stage_piece must not find one
        (void)hipFree(idx->stg_buf);
        idx->stg_buf = nullptr;
        idx->stg_stream = nullptr;
        return set_err(MQ_EHIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(es));
    }
    idx->stg_bytes = total_bytes;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_stage_piece(mq_index *idx, uint64_t at, const uint8_t *src, uint64_t n, uint64_t *ticket) try {
    if (!idx || (!src && n) || !ticket) return set_err(MQ_EINVAL, "bad arguments");
    int rc = use_device(idx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(idx->stg_mu);
    if (!idx->stg_buf) return set_err(MQ_ESTATE, "mq_index_stage_piece before mq_index_stage_begin");
    if (at > idx->stg_bytes || n > idx->stg_bytes - at) return set_err(MQ_EINVAL, "piece outside the staging buffer");
    if (n) HIPCHK(hipMemcpyAsync(idx->stg_buf + at, src, n, hipMemcpyHostToDevice, idx->stg_stream));
    // ticket t <=> stg_events[t], a RECORDED event: the event joins the list only once its record has succeeded (a ticket that indexed an
    // event never recorded would let a wait return at once and a record be indexed before its bytes arrive)
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const hipError_t er = hipEventRecord(ev, idx->stg_stream);
    if (er != hipSuccess) {
        (void)hipEventDestroy(ev);
        return set_err(MQ_EHIP, std::string("hipEventRecord: ") + hipGetErrorString(er));
    }
    idx->stg_events.push_back(ev);
    *ticket = idx->stg_issued++;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_stage_done(mq_index *idx, uint64_t ticket, int wait) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    int rc = use_device(idx);
    if (rc) return rc;
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(idx->stg_mu);
        if (ticket >= idx->stg_issued) return set_err(MQ_EINVAL, "unknown ticket");
        ev = idx->stg_events[(size_t)ticket];
    }
    if (wait) {
        HIPCHK(hipEventSynchronize(ev));
        return 1;
    }
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return 0;
    }
    return set_err(MQ_EHIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int64_t mq_index_add_ref_staged(mq_index *idx, uint32_t ref_id, const char *name, uint64_t at, uint64_t len, uint64_t after_ticket) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    int rc = use_device(idx);
    if (rc) return rc;
    const uint8_t *d_seq = nullptr;
    {
        std::lock_guard<std::mutex> lk(idx->stg_mu);
        if (!idx->stg_buf) return set_err(MQ_ESTATE, "mq_index_add_ref_staged before mq_index_stage_begin");
        if (at > idx->stg_bytes || len > idx->stg_bytes - at) return set_err(MQ_EINVAL, "record outside the staging buffer");
        // the build's kernels run on the null stream: it waits (on the device, not here) for the piece named (pieces complete in issue
        // order, so for every piece up to it), or for every piece issued so far
        if (after_ticket != MQ_STAGE_ALL_ISSUED && after_ticket >= idx->stg_issued) return set_err(MQ_EINVAL, "unknown ticket");
        const uint64_t upto = after_ticket == MQ_STAGE_ALL_ISSUED ? idx->stg_issued : after_ticket + 1;
        if (upto) HIPCHK(hipStreamWaitEvent(0, idx->stg_events[(size_t)upto - 1], 0));
        d_seq = idx->stg_buf + at;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    return add_ref_device_locked(idx, ref_id, name, d_seq, len);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// Slots of the table per inserted k-min-mer (rounded up to a power of two of slots).  The default, 8 (load <= 1/8: 17 GB for a human
// genome), is for a kernel fed from HBM: 1130 Gbases/s against 1114 / 1074 at 4 / 2.  A caller that feeds from files is bound by its
// host side at a thirtieth of that and does better with 2: a quarter of the memory per replica, of the device-to-device copy per clone,
// and of the allocation (fresh device memory can cost 30 ms per GB here) -- the native driver's default.
int mq_index_set_table_factor(mq_index *idx, uint32_t slots_per_kminmer) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    if (slots_per_kminmer < 2 || slots_per_kminmer > 64) return set_err(MQ_EINVAL, "slots per k-min-mer: 2..64");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->finalized || idx->rsv_thread.joinable() || idx->rsv_table) return set_err(MQ_ESTATE, "mq_index_set_table_factor: before mq_index_reserve / mq_index_finalize");
    idx->table_factor = slots_per_kminmer;
    return MQ_OK;
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// DashMap::with_capacity at Index::new (src/index.rs:83 sizes the map for 39,821,990 k-min-mers before the first insert): the table
// for `expected_kminmers` inserted k-min-mers is allocated and cleared by a thread of its own, while the caller reads, uploads and
// seeds the reference -- fresh device memory costs ~30 ms per GB on this platform (tools/alloc_probe.hip: 485 ms for the 17 GB table
// of a human genome), more than every kernel of the build together.  A hint only: mq_index_finalize allocates again when the
// reference turns out to need another size.
int mq_index_reserve(mq_index *idx, uint64_t expected_kminmers) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->finalized) return set_err(MQ_ESTATE, "index already finalized");
    if (idx->rsv_thread.joinable() || idx->rsv_table) return MQ_OK;  // one reservation per index
    const uint64_t nslots = table_slots_for(idx, expected_kminmers);
    idx->rsv_nslots = nslots;
    const int device = idx->device;
    idx->rsv_thread = std::thread([idx, nslots, device]() {
        hipError_t e = hipSetDevice(device);
        void *p = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        if (e == hipSuccess) e = hipMalloc(&p, table_bytes_of(nslots));
        hipStream_t st = nullptr;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);  // not the null stream: the build's kernels run there
        if (e == hipSuccess) e = hipMemsetAsync(p, 0, table_bytes_of(nslots), st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        idx->rsv_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (st) hipStreamDestroy(st);
        if (e != hipSuccess && p) {
            hipFree(p);
            p = nullptr;
        }
        idx->rsv_table = (Bucket *)p;
        idx->rsv_err = (int)e;
    });
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int64_t mq_index_finalize(mq_index *idx) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->finalized) return (int64_t)idx->n_unique;
    int rc = use_device(idx);
    if (rc) return rc;
    const bool timing = getenv("MQ_BUILD_TIMING") != nullptr;  // diagnostic: where the wall time of finalize goes (stderr)
    auto tnow = [&]() {
        if (timing) hipDeviceSynchronize();
        return std::chrono::steady_clock::now();
    };
    auto t_0 = tnow();
    // slots per inserted k-min-mer (power-of-two rounding on top); default 8 => load <= 0.125 (17 GB for a human genome, 6 % of
    // the HBM).  ~85 % of a read's lookups miss, a miss walks to the first empty slot, and every extra step is one more dependent
    // random access of a memory system that sustains ~52 G of them per second (tools/probe_rate.py).  Measured on the CHM13-like
    // bench: factor 2: 926, 4: 1000, 8: 1034, 16: 1044, 32: 1051 Gbases/s.
    const uint64_t nslots = table_slots_for(idx, idx->n_kmm_total);
    rsv_join(idx);
    if (idx->rsv_table && idx->rsv_nslots == nslots && idx->rsv_err == 0) {  // the table mq_index_reserve allocated and cleared
        if (idx->table) HIPCHK(hipFree(idx->table));
        idx->table = idx->rsv_table;
        idx->nslots = nslots;
        idx->rsv_table = nullptr;
        idx->table_alloc_ms = idx->rsv_ms;
    } else {
        if (idx->rsv_table) {
            HIPCHK(hipFree(idx->rsv_table));  // the estimate was off: the table is allocated now, at the size the reference needs
            idx->rsv_table = nullptr;
        }
        rc = alloc_table(idx, nslots);
        if (rc) return rc;
    }
    auto t_1 = tnow();
    unsigned long long *d_acc = nullptr;
    HIPCHK(hipMalloc((void **)&d_acc, 24));
    HIPCHK(hipMemset(d_acc, 0, 24));
    for (auto &c : idx->chunks) {
        if (!c.n) continue;
        const uint32_t nb = (uint32_t)std::min<uint64_t>((c.n + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(insert_kernel, dim3(nb), dim3(256), 0, 0, c.d, c.n, idx->table, nslots - 1, d_acc);
        HIPCHK(hipGetLastError());
    }
    auto t_2 = tnow();
    // Index::get_count (src/index.rs:90-92): keys claimed minus keys that turned dead, counted by the insertions themselves
    unsigned long long acc[3] = {0, 0, 0};
    HIPCHK(hipMemcpy(acc, d_acc, 24, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d_acc));
    idx->n_keys = acc[0];
    idx->n_unique = acc[0] - acc[1];
    auto t_3 = tnow();
    for (auto &c : idx->chunks)
        if (c.d) hipFree(c.d);
    idx->chunks.clear();
    free_build_scratch(idx);
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "mq_index_add_ref_device calls so far: %.2f ms\n", idx->t_add_ms);
        fprintf(stderr, "mq_index_finalize: table alloc + clear %.2f ms (%.1f GB), insert %.2f ms (%llu k-min-mers), read back %.2f ms, free scratch %.2f ms\n", ms(t_0, t_1),
                table_bytes_of(nslots) / 1e9, ms(t_1, t_2), (unsigned long long)idx->n_kmm_total, ms(t_2, t_3), ms(t_3, tnow()));
    }
    // ref_map lengths (src/closures.rs:49), dense by ref id
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    std::vector<uint64_t> lens((size_t)max_id + 1, 0);
    for (auto &kv : idx->refs) lens[kv.first] = kv.second.second;
    HIPCHK(hipMalloc((void **)&idx->d_ref_lens, lens.size() * sizeof(uint64_t)));
    HIPCHK(hipMemcpy(idx->d_ref_lens, lens.data(), lens.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    idx->finalized = true;
    return (int64_t)idx->n_unique;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_get_stats(const mq_index *idx, mq_index_stats *out) try {
    if (!idx || !out) return set_err(MQ_EINVAL, "bad arguments");
    out->n_refs = idx->refs.size();
    out->n_kminmers = idx->n_kmm_total;
    out->n_keys = idx->n_keys;
    out->n_unique = idx->n_unique;
    out->table_slots = idx->nslots;
    out->table_bytes = table_bytes_of(idx->nslots);
    out->slot_bytes = SLOT_BYTES;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

}  // extern "C"

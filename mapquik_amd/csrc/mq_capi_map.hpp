// mq_capi_map.hpp -- C ABI, mapping side: launch sequences, stream-slot contexts (mq_ctx_*), host-buffer and device-resident entry
// points, device-parsed FASTA chunks, PAF formatting, page-locked host memory (part of the one translation unit mq_capi.hip).
#pragma once

struct LaunchOpt {
    mq_kminmer *d_dump = nullptr;
    const uint64_t *d_dump_off = nullptr;
    uint32_t *d_dump_counts = nullptr;
    MatchRec *scratch_override = nullptr;  // overflow redo: worst-case Match scratch on a small grid
    uint32_t cap_override = 0;
    uint32_t grid_override = 0;
    uint32_t f16 = 0;                      // 0 => list_f16(idx)
    const uint32_t *d_lens = nullptr;      // spans form: per-read lengths
    bool instrumented = false;             // mq_map_probe_stats: the launch that counts lookups and probe steps (slower, never timed)
};

// One launch sequence on stream `st` using the context's scratch.  ctx_ensure(c, n, total_bases, f16) must have succeeded.
static int launch_map(mq_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, mq_hit *d_out, hipStream_t st,
                      const LaunchOpt &o = LaunchOpt()) {
    mq_index *idx = c->idx;
    if (n == 0) return MQ_OK;
    HIPCHK(hipMemsetAsync(c->d_counter, 0, 256, st));
    HIPCHK(hipEventRecord(c->ev0, st));
    SplitArgs A;
    A.bases = d_bases;
    A.offsets = d_offsets;
    A.lens = o.d_lens;
    A.n = n;
    A.P = idx->dp;
    A.mz_hash = c->mz_hash;
    A.mz_pos = c->mz_pos;
    A.mz_last = c->mz_last;
    A.mz_count = c->mz_count;
    A.mz_base = c->mz_base;
    A.pool_base = c->pool_base;
    A.pool_cap = c->pool_cap;
    A.f16 = o.f16 ? o.f16 : list_f16(idx);
    A.slack = LIST_SLACK;
    A.queue = c->queue;
    A.counters = c->d_counter;
    A.force_general = idx->force_general ? 1u : 0u;
    A.table = idx->table;
    A.mask = idx->nslots - 1;
    A.ref_lens = idx->d_ref_lens;
    A.scratch_all = o.scratch_override ? o.scratch_override : c->scratch;
    A.cap_matches = o.scratch_override ? o.cap_override : idx->cap_matches;
    A.out = d_out;
    A.dump = o.d_dump;
    A.dump_off = o.d_dump_off;
    A.dump_counts = o.d_dump_counts;
    A.stats64 = reinterpret_cast<unsigned long long *>(c->d_counter + 8);
    A.work = c->work;
    A.heavy_first = idx->heavy_first ? 1u : 0u;
    if (!idx->split) {
        if (n > WORK_ID_MASK) return set_err(MQ_EINVAL, "more than 2^30 - 1 reads in one batch");
        hipLaunchKernelGGL(order_reads_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, A);  // the launch order (and map_kernel's work descriptors)
        uint32_t grid = std::min<uint32_t>(idx->grid_fused, (n + MAP_WAVES - 1) / MAP_WAVES);
        if (o.grid_override) grid = std::min(grid, o.grid_override);
        const dim3 blk(64 * MAP_WAVES);
        if (idx->dp.variant) {  // a seeding variant other than the frozen reading: the instantiation built with the variants
            if (o.instrumented) hipLaunchKernelGGL((map_kernel<64, true, true>), dim3(grid), blk, 0, st, A);
            else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_kernel<4, false, true>), dim3(grid), blk, 0, st, A);
            else hipLaunchKernelGGL((map_kernel<64, false, true>), dim3(grid), blk, 0, st, A);
        } else if (o.instrumented) hipLaunchKernelGGL((map_kernel<64, true>), dim3(grid), blk, 0, st, A);
        else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_kernel<4, false>), dim3(grid), blk, 0, st, A);
        else hipLaunchKernelGGL((map_kernel<64, false>), dim3(grid), blk, 0, st, A);
        HIPCHK(hipGetLastError());
        // the reads the fast seeder declined (queue length on the device: map_kernel's grid, its waves leave at once when there are none)
        const uint32_t gd = grid;
        if (idx->dp.variant) {
            if (o.instrumented) hipLaunchKernelGGL((map_declined_kernel<64, true, true>), dim3(gd), blk, 0, st, A);
            else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_declined_kernel<4, false, true>), dim3(gd), blk, 0, st, A);
            else hipLaunchKernelGGL((map_declined_kernel<64, false, true>), dim3(gd), blk, 0, st, A);
        } else if (o.instrumented) hipLaunchKernelGGL((map_declined_kernel<64, true>), dim3(gd), blk, 0, st, A);
        else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_declined_kernel<4, false>), dim3(gd), blk, 0, st, A);
        else hipLaunchKernelGGL((map_declined_kernel<64, false>), dim3(gd), blk, 0, st, A);
        HIPCHK(hipGetLastError());
    } else {
        const uint32_t gs = std::min<uint32_t>(idx->grid_seed, (n + SEED_WAVES - 1) / SEED_WAVES);
        const char *ss = getenv("MQ_SEED_STOP");  // diagnostic: stage attribution by truncation (results are NOT valid)
        const int stop = ss ? atoi(ss) : 0;
        if (stop == 1) hipLaunchKernelGGL(seed_reads_kernel<1>, dim3(gs), dim3(64 * SEED_WAVES), 0, st, A);
        else if (stop == 2) hipLaunchKernelGGL(seed_reads_kernel<2>, dim3(gs), dim3(64 * SEED_WAVES), 0, st, A);
        else hipLaunchKernelGGL(seed_reads_kernel<0>, dim3(gs), dim3(64 * SEED_WAVES), 0, st, A);
        HIPCHK(hipGetLastError());
        // the reads the fast seeder declined: the queue length lives on the device, so the grid is fixed and waves that find
        // the queue empty leave at once
        const uint32_t gg = std::min<uint32_t>((uint32_t)idx->n_cu * 8u, n);
        hipLaunchKernelGGL(seed_general_kernel, dim3(gg), dim3(64), 0, st, A);
        HIPCHK(hipGetLastError());
        uint32_t gm = std::min<uint32_t>(idx->grid_map, (n + ML_WAVES - 1) / ML_WAVES);
        if (o.grid_override) gm = std::min(gm, o.grid_override);
        const dim3 blk(64 * ML_WAVES);
        if (o.instrumented) hipLaunchKernelGGL((map_lists_kernel<64, true>), dim3(gm), blk, 0, st, A);
        else if (idx->chain_chunk == 4) hipLaunchKernelGGL((map_lists_kernel<4, false>), dim3(gm), blk, 0, st, A);
        else hipLaunchKernelGGL((map_lists_kernel<64, false>), dim3(gm), blk, 0, st, A);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev1, st));
    c->ev_valid = true;
    return MQ_OK;
}

// Reads that came back MQ_HIT_OVERFLOW (more Match runs than the per-wave scratch holds, or a minimizer list denser than its
// region): map those again on the GPU with worst-case scratch and list regions on a small grid.  Never a CPU path.
static int redo_overflow(mq_ctx *c, const uint8_t *bases, const uint64_t *offsets, const uint32_t *lens, uint32_t n, mq_hit *out) {
    mq_index *idx = c->idx;
    std::vector<uint32_t> redo;
    for (uint32_t i = 0; i < n; ++i)
        if (out[i].status == MQ_HIT_OVERFLOW) redo.push_back(i);
    if (redo.empty()) return MQ_OK;
    uint64_t sub_max = 0;
    std::vector<uint64_t> so(redo.size() + 1, 0);
    for (size_t j = 0; j < redo.size(); ++j) {
        const uint64_t L = lens ? (uint64_t)lens[redo[j]] : offsets[redo[j] + 1] - offsets[redo[j]];
        so[j + 1] = so[j] + L;
        sub_max = std::max(sub_max, L);
    }
    const uint64_t sub_total = so.back();
    std::vector<uint8_t> sb(sub_total ? sub_total : 1);
    for (size_t j = 0; j < redo.size(); ++j) memcpy(sb.data() + so[j], bases + offsets[redo[j]], (size_t)(so[j + 1] - so[j]));
    const uint32_t cap = (uint32_t)std::max<uint64_t>(sub_max, 1);  // a read cannot have more runs than bases
    const uint32_t waves = std::max(MAP_WAVES, ML_WAVES);
    const uint32_t rgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(idx->grid_fused, idx->grid_map), (1ull << 30) / ((uint64_t)cap * sizeof(MatchRec) * waves)));
    int rc = ctx_ensure(c, (uint32_t)redo.size(), sub_total, 65536u);
    if (rc) return rc;
    MatchRec *big = nullptr;
    uint8_t *d_sb = nullptr;
    uint64_t *d_so = nullptr;
    mq_hit *d_sh = nullptr;
    hipError_t e = hipMalloc((void **)&big, (size_t)rgrid * waves * cap * sizeof(MatchRec));
    if (e == hipSuccess) e = hipMalloc((void **)&d_sb, sub_total + 1);
    if (e == hipSuccess) e = hipMalloc((void **)&d_so, so.size() * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sh, redo.size() * sizeof(mq_hit));
    if (e == hipSuccess && sub_total) e = hipMemcpyAsync(d_sb, sb.data(), sub_total, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_so, so.data(), so.size() * 8, hipMemcpyHostToDevice, c->stream);
    int rrc = MQ_OK;
    std::vector<mq_hit> sh(redo.size());
    if (e == hipSuccess) {
        LaunchOpt o;
        o.scratch_override = big;
        o.cap_override = cap;
        o.grid_override = rgrid;
        o.f16 = 65536u;
        rrc = launch_map(c, d_sb, d_so, (uint32_t)redo.size(), d_sh, c->stream, o);
    }
    if (e == hipSuccess && rrc == MQ_OK) e = hipMemcpyAsync(sh.data(), d_sh, redo.size() * sizeof(mq_hit), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    else hipStreamSynchronize(c->stream);
    hipFree(big);
    hipFree(d_sb);
    hipFree(d_so);
    hipFree(d_sh);
    if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, std::string("overflow retry: ") + hipGetErrorString(e));
    if (rrc) return rrc;
    for (size_t j = 0; j < redo.size(); ++j) out[redo[j]] = sh[j];
    return MQ_OK;
}

// host buffers -> device staging -> launch sequence -> page-locked hits, all asynchronous on the context's stream.
// lens == nullptr: offsets has n + 1 entries and read i is bases[offsets[i], offsets[i+1]).  lens != nullptr (spans form): the
// whole buffer bases[0, buf_bytes) goes to the device and read i is bases[offsets[i], offsets[i] + lens[i]) (n offsets).
static int ctx_submit(mq_ctx *c, const uint8_t *bases, uint64_t buf_bytes, const uint64_t *offsets, const uint32_t *lens, uint32_t n,
                      mq_hit *out) {
    mq_index *idx = c->idx;
    if (c->pending || c->fx_pending) return set_err(MQ_ESTATE, "context has a submitted batch: call mq_ctx_wait / mq_ctx_wait_fasta first");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (n == 0) return MQ_OK;
    int rc = use_device(idx);
    if (rc) return rc;
    if ((rc = grow_pinned(c->h_off, c->h_off_cap, (uint64_t)n + 1))) return rc;
    if ((rc = grow_pinned(c->h_out, c->h_out_cap, (uint64_t)n))) return rc;
    uint64_t total, first;
    if (!lens) {
        first = offsets[0];
        total = offsets[n] - offsets[0];
        for (uint32_t i = 0; i < n; ++i) {
            if (offsets[i + 1] < offsets[i]) return set_err(MQ_EINVAL, "offsets must be non-decreasing");
            if (offsets[i + 1] - offsets[i] >= (1ull << 32)) return set_err(MQ_EINVAL, "sequence length must be < 2^32");
            c->h_off[i] = offsets[i] - first;
        }
    } else {
        first = 0;
        total = buf_bytes;
        uint64_t prev_end = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (offsets[i] < prev_end || offsets[i] + lens[i] > buf_bytes) return set_err(MQ_EINVAL, "spans must be in order, disjoint and inside the buffer");
            prev_end = offsets[i] + lens[i];
            c->h_off[i] = offsets[i];
        }
    }
    c->h_off[n] = total;
    if ((rc = ctx_ensure(c, n, total, list_f16(idx)))) return rc;
    if ((rc = grow(c->st_bases, c->st_bases_cap, total + 64))) return rc;
    if ((rc = grow(c->st_off, c->st_off_cap, (uint64_t)n + 1))) return rc;
    if ((rc = grow(c->st_out, c->st_out_cap, (uint64_t)n))) return rc;
    if (lens && (rc = grow(c->st_lens, c->st_lens_cap, (uint64_t)n))) return rc;
    if (total) HIPCHK(hipMemcpyAsync(c->st_bases, bases + first, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->st_off, c->h_off, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    LaunchOpt o;
    if (lens) {
        HIPCHK(hipMemcpyAsync(c->st_lens, lens, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        o.d_lens = c->st_lens;
    }
    rc = launch_map(c, c->st_bases, c->st_off, n, c->st_out, c->stream, o);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->h_out, c->st_out, (size_t)n * sizeof(mq_hit), hipMemcpyDeviceToHost, c->stream));
    c->pending = true;
    c->p_bases = bases;
    c->p_offsets = offsets;
    c->p_lens = lens;
    c->p_n = n;
    c->p_out = out;
    return MQ_OK;
}

static int ctx_wait(mq_ctx *c) {
    if (!c->pending) return MQ_OK;
    c->pending = false;
    int rc = use_device(c->idx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(c->p_out, c->h_out, (size_t)c->p_n * sizeof(mq_hit));
    return redo_overflow(c, c->p_bases, c->p_offsets, c->p_lens, c->p_n, c->p_out);
}

static int ctx_map_device(mq_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases, mq_hit *d_out,
                          hipStream_t st, bool instrumented = false) {
    mq_index *idx = c->idx;
    if (n && (!d_offsets || !d_out)) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (c->pending || c->fx_pending) return set_err(MQ_ESTATE, "context has a submitted batch: call mq_ctx_wait / mq_ctx_wait_fasta first");
    int rc = use_device(idx);
    if (rc) return rc;
    if ((rc = ctx_ensure(c, n, total_bases, list_f16(idx)))) return rc;
    LaunchOpt o;
    o.instrumented = instrumented;
    return launch_map(c, d_bases, d_offsets, n, d_out, st, o);
}

// ---- device-parsed FASTA chunks: the raw bytes go to the device, the scan kernels of mq_fastx.hpp find the records, map_kernel takes
// the spans from device memory.  Two steps, so that the host learns the number of records without standing in the stream's way:
// submit = copy + scan + the scan's result words back (all asynchronous); wait = read them, launch the map kernels, hits and line ends
// back.  The copy of the NEXT chunk (another context, another stream) runs meanwhile: the link stays busy.
static uint32_t fx_line_cap(uint64_t bytes) { return (uint32_t)std::min<uint64_t>(bytes / 16 + 4096, 1u << 28); }

static int ctx_submit_fasta(mq_ctx *c, const uint8_t *buf, uint64_t begin, uint64_t bytes, uint32_t format) {
    mq_index *idx = c->idx;
    if (c->pending || c->fx_pending) return set_err(MQ_ESTATE, "context has a submitted batch: call mq_ctx_wait first");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (bytes >= (1ull << 32) || begin > bytes) return set_err(MQ_EINVAL, "a chunk must be smaller than 4 GB");
    if (format != MQ_FASTX_FASTA && format != MQ_FASTX_FASTQ) return set_err(MQ_EINVAL, "format must be MQ_FASTX_FASTA or MQ_FASTX_FASTQ");
    const uint32_t lpr = format == MQ_FASTX_FASTQ ? 4u : 2u;  // lines per record
    int rc = use_device(idx);
    if (rc) return rc;
    const uint32_t n_tiles = (uint32_t)((bytes + FX_TILE - 1) / FX_TILE);
    const uint32_t cap = fx_line_cap(bytes);
    if ((rc = grow(c->st_bases, c->st_bases_cap, bytes + 64))) return rc;
    if ((rc = grow(c->fx_tile_counts, c->fx_tile_counts_cap, (uint64_t)n_tiles + 1))) return rc;
    if ((rc = grow(c->fx_tile_off, c->fx_tile_off_cap, (uint64_t)n_tiles + 1))) return rc;
    if ((rc = grow(c->fx_nl, c->fx_nl_cap, cap))) return rc;
    if ((rc = grow(c->st_off, c->st_off_cap, (uint64_t)cap / 2 + 1))) return rc;
    if ((rc = grow(c->st_lens, c->st_lens_cap, (uint64_t)cap / 2 + 1))) return rc;
    if (!c->fx_info) HIPCHK(hipMalloc((void **)&c->fx_info, 16));
    if (!c->h_fx_info) HIPCHK(hipHostMalloc((void **)&c->h_fx_info, 16, hipHostMallocDefault));
    hipStream_t st = c->stream;
    if (bytes) {
        // The bytes behind the last page boundary (< 4 KB) go through a page-locked buffer of the context: a caller that page-locks the
        // whole pages of its pieces (the feeder, on a mapped file: mq_host_register) gets an asynchronous copy for all the rest, and the
        // last partial page -- which may belong to a range somebody else locks and releases -- is never the source of a DMA.
        const uintptr_t end_addr = (uintptr_t)buf + bytes;
        const uintptr_t cut = end_addr & ~(uintptr_t)4095;
        uint64_t main_len = cut > (uintptr_t)buf ? (uint64_t)(cut - (uintptr_t)buf) : 0;
        const uint64_t tail_len = bytes - main_len;
        if (!c->h_fx_tail) HIPCHK(hipHostMalloc((void **)&c->h_fx_tail, 4096, hipHostMallocDefault));
        if (main_len) HIPCHK(hipMemcpyAsync(c->st_bases, buf, main_len, hipMemcpyHostToDevice, st));
        if (tail_len) {
            memcpy(c->h_fx_tail, buf + main_len, tail_len);
            HIPCHK(hipMemcpyAsync(c->st_bases + main_len, c->h_fx_tail, tail_len, hipMemcpyHostToDevice, st));
        }
    }
    const uint32_t b = (uint32_t)begin, e = (uint32_t)bytes;
    const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>((n_tiles + 3) / 4, (uint32_t)idx->n_cu * 8u));
    hipLaunchKernelGGL(count_newlines_kernel, dim3(grid), dim3(256), 0, st, c->st_bases, b, e, n_tiles, c->fx_tile_counts);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, st, c->st_bases, b, e, c->fx_tile_counts, n_tiles, c->fx_tile_off, c->fx_nl, cap, c->fx_info, lpr);
    hipLaunchKernelGGL(list_newlines_kernel, dim3(grid), dim3(256), 0, st, c->st_bases, b, e, n_tiles, c->fx_tile_off, c->fx_nl, cap);
    const dim3 sgrid(std::max<uint32_t>(1, std::min<uint32_t>(cap / 2 / 256 + 1, (uint32_t)idx->n_cu * 4u)));
    if (format == MQ_FASTX_FASTQ)
        hipLaunchKernelGGL(fastq_spans_kernel, sgrid, dim3(256), 0, st, c->st_bases, b, e, c->fx_nl, c->fx_info, reinterpret_cast<unsigned long long *>(c->st_off), c->st_lens, cap / 2);
    else
        hipLaunchKernelGGL(fasta_spans_kernel, sgrid, dim3(256), 0, st, c->st_bases, b, e, c->fx_nl, c->fx_info, reinterpret_cast<unsigned long long *>(c->st_off), c->st_lens, cap / 2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_fx_info, c->fx_info, 16, hipMemcpyDeviceToHost, st));
    c->fx_pending = true;
    c->fx_lpr = lpr;
    c->fx_buf = buf;
    c->fx_begin = b;
    c->fx_bytes = e;
    return MQ_OK;
}

static int ctx_wait_fasta(mq_ctx *c, uint32_t *n_reads, const uint32_t **line_ends, uint32_t *n_lines, const mq_hit **hits, uint32_t *flags) {
    if (!c->fx_pending) return set_err(MQ_ESTATE, "no FASTA chunk submitted on this context");
    c->fx_pending = false;
    mq_index *idx = c->idx;
    int rc = use_device(idx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t lines = c->h_fx_info[0], n = c->h_fx_info[1];
    *flags = c->h_fx_info[2];
    *n_reads = 0;
    *n_lines = 0;
    *line_ends = nullptr;
    *hits = nullptr;
    if (*flags & FX_IRREGULAR) return MQ_OK;  // not "header line, sequence line" all through: the caller parses this chunk on the host
    if (n == 0) return MQ_OK;
    if ((rc = grow_pinned(c->h_out, c->h_out_cap, (uint64_t)n))) return rc;
    if ((rc = grow_pinned(c->h_fx_nl, c->h_fx_nl_cap, (uint64_t)lines))) return rc;
    if ((rc = grow(c->st_out, c->st_out_cap, (uint64_t)n))) return rc;
    if ((rc = ctx_ensure(c, n, c->fx_bytes, list_f16(idx)))) return rc;
    LaunchOpt o;
    o.d_lens = c->st_lens;
    if ((rc = launch_map(c, c->st_bases, c->st_off, n, c->st_out, c->stream, o))) return rc;
    HIPCHK(hipMemcpyAsync(c->h_out, c->st_out, (size_t)n * sizeof(mq_hit), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_fx_nl, c->fx_nl, (size_t)lines * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    bool over = false;
    for (uint32_t i = 0; i < n && !over; ++i) over = c->h_out[i].status == MQ_HIT_OVERFLOW;
    if (over) {  // the rare reads with more Match runs / denser lists than the scratch holds: again with room, from the host's copy of the chunk
        std::vector<uint64_t> offs(n);
        std::vector<uint32_t> lens(n);
        const uint32_t lpr = c->fx_lpr;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t ss = c->h_fx_nl[lpr * i] + 1;
            uint32_t e = c->h_fx_nl[lpr * i + 1];
            if (e > ss && c->fx_buf[e - 1] == '\r') --e;
            offs[i] = ss;
            lens[i] = e - ss;
        }
        if ((rc = redo_overflow(c, c->fx_buf, offs.data(), lens.data(), n, c->h_out))) return rc;
    }
    *n_reads = n;
    *n_lines = lines;
    *line_ends = c->h_fx_nl;
    *hits = c->h_out;
    return MQ_OK;
}

extern "C" {

int mq_ctx_submit_fasta(mq_ctx *ctx, const uint8_t *buf, uint64_t begin, uint64_t bytes) try {
    if (!ctx || (bytes && !buf)) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_submit_fasta(ctx, buf, begin, bytes, MQ_FASTX_FASTA);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_submit_fastx(mq_ctx *ctx, const uint8_t *buf, uint64_t begin, uint64_t bytes, uint32_t format) try {
    if (!ctx || (bytes && !buf)) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_submit_fasta(ctx, buf, begin, bytes, format);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_wait_fasta(mq_ctx *ctx, uint32_t *n_reads, const uint32_t **line_ends, uint32_t *n_lines, const mq_hit **hits, uint32_t *flags) try {
    if (!ctx || !n_reads || !line_ends || !n_lines || !hits || !flags) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_wait_fasta(ctx, n_reads, line_ends, n_lines, hits, flags);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

mq_ctx *mq_ctx_new(mq_index *idx) try {
    if (!idx) {
        set_err(MQ_EINVAL, "idx is NULL");
        return nullptr;
    }
    return ctx_create(idx);
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

void mq_ctx_free(mq_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->idx->device);
    ctx_release(ctx);
}

int mq_ctx_submit(mq_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) try {
    if (!ctx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_submit(ctx, bases, 0, offsets, nullptr, n, out);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_submit_spans(mq_ctx *ctx, const uint8_t *buf, uint64_t buf_bytes, const uint64_t *starts, const uint32_t *lens, uint32_t n,
                        mq_hit *out) try {
    if (!ctx || (n && (!buf || !starts || !lens || !out))) return set_err(MQ_EINVAL, "bad arguments");
    return ctx_submit(ctx, buf, buf_bytes, starts, lens, n, out);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_reserve(mq_ctx *ctx, uint32_t n_reads, uint64_t total_bytes) try {
    if (!ctx) return set_err(MQ_EINVAL, "ctx is NULL");
    mq_ctx *c = ctx;
    int rc = use_device(c->idx);
    if (rc) return rc;
    if ((rc = grow_pinned(c->h_off, c->h_off_cap, (uint64_t)n_reads + 1))) return rc;
    if ((rc = grow_pinned(c->h_out, c->h_out_cap, (uint64_t)n_reads))) return rc;
    if ((rc = ctx_ensure(c, n_reads, total_bytes, list_f16(c->idx)))) return rc;
    if ((rc = grow(c->st_bases, c->st_bases_cap, total_bytes + 64))) return rc;
    if ((rc = grow(c->st_off, c->st_off_cap, (uint64_t)n_reads + 1))) return rc;
    if ((rc = grow(c->st_out, c->st_out_cap, (uint64_t)n_reads))) return rc;
    return grow(c->st_lens, c->st_lens_cap, (uint64_t)n_reads);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_wait(mq_ctx *ctx) try {
    if (!ctx) return set_err(MQ_EINVAL, "ctx is NULL");
    return ctx_wait(ctx);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_map_batch(mq_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) try {
    if (!ctx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    int rc = ctx_submit(ctx, bases, 0, offsets, nullptr, n, out);
    if (rc) return rc;
    return ctx_wait(ctx);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_ctx_map_batch_device(mq_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases,
                            mq_hit *d_out, void *stream) try {
    if (!ctx) return set_err(MQ_EINVAL, "ctx is NULL");
    return ctx_map_device(ctx, d_bases, d_offsets, n, total_bases, d_out, (hipStream_t)stream);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_batch_device(mq_index *idx, const uint8_t *d_bases, const uint64_t *d_offsets, uint32_t n, uint64_t total_bases,
                        mq_hit *d_out, void *stream) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    return ctx_map_device(idx->def_ctx, d_bases, d_offsets, n, total_bases, d_out, (hipStream_t)stream);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, mq_hit *out) try {
    if (!idx || (n && (!offsets || !out))) return set_err(MQ_EINVAL, "bad arguments");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = ctx_submit(idx->def_ctx, bases, 0, offsets, nullptr, n, out);
    if (rc) return rc;
    return ctx_wait(idx->def_ctx);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_kminmers_batch(mq_index *idx, const uint8_t *bases, const uint64_t *offsets, uint32_t n, const uint64_t *kmm_offsets,
                      mq_kminmer *out, uint32_t *counts) try {
    if (!idx || (n && (!offsets || !kmm_offsets || !counts))) return set_err(MQ_EINVAL, "bad arguments");
    if (n == 0) return MQ_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    const uint64_t total = offsets[n] - offsets[0];
    const uint64_t ktotal = kmm_offsets[n] - kmm_offsets[0];
    for (uint32_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] >= (1ull << 32)) return set_err(MQ_EINVAL, "bad offsets / sequence length must be < 2^32");
    // parity/debug entry point: list regions sized for the worst case (one minimizer per base), so no sequence overflows
    rc = ctx_ensure(idx->def_ctx, n, total, 65536u);
    if (rc) return rc;
    uint8_t *d_b = nullptr;
    uint64_t *d_o = nullptr, *d_ko = nullptr;
    mq_kminmer *d_k = nullptr;
    uint32_t *d_c = nullptr;
    mq_hit *d_h = nullptr;
    uint64_t *d_zero_lens = nullptr;
    auto cleanup = [&]() {
        hipFree(d_b); hipFree(d_o); hipFree(d_ko); hipFree(d_k); hipFree(d_c); hipFree(d_h); hipFree(d_zero_lens);
    };
    std::vector<uint64_t> rel((size_t)n + 1), krel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) {
        rel[i] = offsets[i] - offsets[0];
        krel[i] = kmm_offsets[i] - kmm_offsets[0];
    }
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipMalloc((void **)&d_b, total + 1));
    ok(hipMalloc((void **)&d_o, ((size_t)n + 1) * 8));
    ok(hipMalloc((void **)&d_ko, ((size_t)n + 1) * 8));
    ok(hipMalloc((void **)&d_k, (ktotal + 1) * sizeof(mq_kminmer)));
    ok(hipMalloc((void **)&d_c, (size_t)n * 4));
    ok(hipMalloc((void **)&d_h, (size_t)n * sizeof(mq_hit)));
    if (e == hipSuccess && total) ok(hipMemcpy(d_b, bases + offsets[0], total, hipMemcpyHostToDevice));
    if (e == hipSuccess) ok(hipMemcpy(d_o, rel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    if (e == hipSuccess) ok(hipMemcpy(d_ko, krel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    if (e != hipSuccess) {
        cleanup();
        return set_err(MQ_EHIP, std::string("mq_kminmers_batch setup: ") + hipGetErrorString(e));
    }
    // before finalize there is no ref table: the 1-slot empty table never hits, so ref_lens is never read
    {
        LaunchOpt o;
        o.d_dump = d_k;
        o.d_dump_off = d_ko;
        o.d_dump_counts = d_c;
        o.f16 = 65536u;
        rc = launch_map(idx->def_ctx, d_b, d_o, n, d_h, 0, o);
    }
    if (rc) {
        cleanup();
        return rc;
    }
    ok(hipMemcpy(counts, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (e == hipSuccess && ktotal && out) ok(hipMemcpy(out + kmm_offsets[0], d_k, ktotal * sizeof(mq_kminmer), hipMemcpyDeviceToHost));
    cleanup();
    if (e != hipSuccess) return set_err(MQ_EHIP, std::string("mq_kminmers_batch copy-out: ") + hipGetErrorString(e));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_lookup(mq_index *idx, const uint64_t *hashes, uint32_t n, uint8_t *found, mq_kminmer *entries, uint32_t *ref_ids) try {
    if (!idx || (n && (!hashes || !found || !entries || !ref_ids))) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    if (n == 0) return MQ_OK;
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    uint64_t *d_k = nullptr;
    uint8_t *d_f = nullptr;
    mq_kminmer *d_e = nullptr;
    uint32_t *d_r = nullptr;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    ok(hipMalloc((void **)&d_k, (size_t)n * 8));
    ok(hipMalloc((void **)&d_f, (size_t)n));
    ok(hipMalloc((void **)&d_e, (size_t)n * sizeof(mq_kminmer)));
    ok(hipMalloc((void **)&d_r, (size_t)n * 4));
    if (e == hipSuccess) ok(hipMemcpy(d_k, hashes, (size_t)n * 8, hipMemcpyHostToDevice));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, idx->table, idx->nslots - 1, d_k, n, d_f, d_e, d_r);
        ok(hipGetLastError());
    }
    if (e == hipSuccess) ok(hipMemcpy(found, d_f, (size_t)n, hipMemcpyDeviceToHost));
    if (e == hipSuccess) ok(hipMemcpy(entries, d_e, (size_t)n * sizeof(mq_kminmer), hipMemcpyDeviceToHost));
    if (e == hipSuccess) ok(hipMemcpy(ref_ids, d_r, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(d_k); hipFree(d_f); hipFree(d_e); hipFree(d_r);
    if (e != hipSuccess) return set_err(MQ_EHIP, std::string("mq_index_lookup: ") + hipGetErrorString(e));
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_format_paf(const mq_index *idx, const char *q_id, uint64_t q_len, const mq_hit *hit, char *buf, size_t cap) try {
    if (!idx || !q_id || !hit || !buf) return set_err(MQ_EINVAL, "bad arguments");
    if (hit->status != MQ_HIT_MAPPED) return set_err(MQ_EINVAL, "hit is not mapped: the reference writes no line");
    auto it = idx->refs.find(hit->ref_id);
    if (it == idx->refs.end()) return set_err(MQ_EINVAL, "unknown ref_id in hit");
    const unsigned long long r_len = it->second.second;
    // src/mers.rs:181: column 11 repeats r_len, column 10 is the score
    const unsigned long long qs = ((unsigned long long)hit->q_start_hi << 32) | hit->q_start, qe = ((unsigned long long)hit->q_end_hi << 32) | hit->q_end;
    int w = snprintf(buf, cap, "%s\t%llu\t%llu\t%llu\t%s\t%s\t%llu\t%u\t%u\t%u\t%llu\t%u", q_id, (unsigned long long)q_len, qs, qe,
                     hit->rc ? "-" : "+", it->second.first.c_str(), r_len, hit->r_start, hit->r_end, hit->score, r_len,
                     hit->mapq);
    return w;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// Page-locked host memory.  hipHostMalloc pins at ~4 GB/s on this platform (and hipHostFree costs another 0.14 s per GB), which
// made the feeder's chunk pool the start-up cost of the read phase; an anonymous mapping backed by transparent huge pages,
// touched and then registered, is page-locked at ~15 GB/s and copies to the device at the full PCIe rate
// (tools/pin_rate.hip, profiles/r03_pin_rate.txt).  Falls back to hipHostMalloc when the mapping or the registration fails.
namespace {
std::mutex g_host_mu;
std::map<void *, std::pair<size_t, bool>> g_host_allocs;  // pointer -> (mapped bytes, true: mmap + hipHostRegister)
}  // namespace

void *mq_host_alloc(size_t bytes) {
    if (!bytes) bytes = 1;
    const size_t huge = 2u << 20;
    const size_t mapped = (bytes + huge - 1) / huge * huge;
    void *p = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p != MAP_FAILED) {
        madvise(p, mapped, MADV_HUGEPAGE);
        for (size_t o = 0; o < mapped; o += 4096) ((volatile uint8_t *)p)[o] = 0;  // fault the pages in (2 MB at a time under THP)
        if (hipHostRegister(p, mapped, hipHostRegisterPortable) == hipSuccess) {  // portable: a feeder's chunk goes to whichever GPU's worker takes it
            std::lock_guard<std::mutex> lk(g_host_mu);
            g_host_allocs[p] = std::make_pair(mapped, true);
            return p;
        }
        (void)hipGetLastError();
        munmap(p, mapped);
    }
    p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        set_err(e == hipErrorOutOfMemory ? MQ_ENOMEM : MQ_EHIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_allocs[p] = std::make_pair(bytes, false);
    return p;
}

// Page-lock a range the caller owns (a slice of a mapped file: the copy to the device then runs by DMA straight out of the page cache,
// asynchronously, at the link's rate -- tools/file_h2d.hip).  ptr and bytes whole pages.  0 on success.
int mq_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return set_err(MQ_EINVAL, "bad arguments");
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_err(MQ_EHIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
    }
    return MQ_OK;
}
int mq_host_unregister(void *ptr) {
    if (!ptr) return MQ_OK;
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_err(MQ_EHIP, std::string("hipHostUnregister: ") + hipGetErrorString(e));
    }
    return MQ_OK;
}

void mq_host_free(void *p) {
    if (!p) return;
    std::pair<size_t, bool> info(0, false);
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_allocs.find(p);
        if (it == g_host_allocs.end()) return;  // not ours
        info = it->second;
        g_host_allocs.erase(it);
    }
    if (info.second) {
        hipHostUnregister(p);
        munmap(p, info.first);
    } else {
        hipHostFree(p);
    }
}

}  // extern "C"

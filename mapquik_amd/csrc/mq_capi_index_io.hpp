// mq_capi_index_io.hpp -- C ABI: the on-disk index (mq_index_save / mq_index_load), replicas (mq_index_clone), reference info (part of the one
// translation unit mq_capi.hip).
#pragma once

extern "C" {

// On-disk index (the reference has none and rebuilds on every run, src/closures.rs:24-94): header, parameters, reference table,
// then the OCCUPIED slots only (32 bytes each: ~1.5 GB for a human genome instead of the 17 GB table at load 1/8); mq_index_load
// scatters them into a fresh table on the device.  Little-endian, this library's layout (MQ_INDEX_MAGIC names the version).
static const char MQ_INDEX_MAGIC[8] = {'M', 'Q', 'H', 'I', 'P', 'I', 'X', '2'};
constexpr size_t IX_IO_CHUNK = 64u << 20;  // bytes per page-locked transfer buffer (two of them: the copy overlaps the file I/O)

static bool write_all(int fd, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    while (n) {
        const ssize_t w = ::write(fd, b, n);
        if (w <= 0) return false;
        b += w;
        n -= (size_t)w;
    }
    return true;
}
static bool read_all(int fd, void *p, size_t n) {
    uint8_t *b = (uint8_t *)p;
    while (n) {
        const ssize_t r = ::read(fd, b, n);
        if (r <= 0) return false;
        b += r;
        n -= (size_t)r;
    }
    return true;
}

int mq_index_save(const mq_index *idx, const char *path) try {
    if (!idx || !path) return set_err(MQ_EINVAL, "bad arguments");
    if (!idx->finalized) return set_err(MQ_ESTATE, "index not finalized");
    int rc = use_device(idx);
    if (rc) return rc;
    // occupied slots, packed on the device
    const uint64_t n_occ = idx->n_keys;
    SavedSlot *d_pack = nullptr;
    unsigned long long *d_cur = nullptr;
    uint8_t *h_buf[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int fd = -1;
    auto cleanup = [&]() {
        hipFree(d_pack);
        hipFree(d_cur);
        for (int i = 0; i < 2; ++i) {
            if (h_buf[i]) hipHostFree(h_buf[i]);
            if (ev[i]) hipEventDestroy(ev[i]);
        }
        if (st) hipStreamDestroy(st);
        if (fd >= 0) ::close(fd);
    };
    auto fail = [&](int code, const std::string &msg) {
        cleanup();
        return set_err(code, msg);
    };
    if (hipMalloc((void **)&d_pack, (size_t)(n_occ + 1) * sizeof(SavedSlot)) != hipSuccess || hipMalloc((void **)&d_cur, 8) != hipSuccess ||
        hipMemset(d_cur, 0, 8) != hipSuccess)
        return fail(MQ_ENOMEM, "mq_index_save: no device memory for the packed slots");
    const uint64_t nb1 = idx->nslots / 2 + 1;
    hipLaunchKernelGGL(pack_slots_kernel, dim3((uint32_t)std::min<uint64_t>((2 * nb1 + 255) / 256, 1u << 16)), dim3(256), 0, 0, idx->table, nb1, d_pack,
                       d_cur, n_occ);
    unsigned long long packed = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpy(&packed, d_cur, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(MQ_EHIP, "mq_index_save: packing the table failed");
    if (packed != n_occ) return fail(MQ_ESTATE, "mq_index_save: the table holds another number of keys than the index records (internal error)");
    fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(MQ_EINVAL, std::string("cannot open for writing: ") + path);
    bool ok = write_all(fd, MQ_INDEX_MAGIC, 8);
    const uint64_t hdr[6] = {sizeof(SavedSlot), idx->nslots, idx->n_kmm_total, idx->n_keys, idx->n_unique, (uint64_t)idx->refs.size()};
    ok = ok && write_all(fd, &idx->params, sizeof(mq_params)) && write_all(fd, hdr, sizeof(hdr));
    for (auto &kv : idx->refs) {
        const uint32_t id = kv.first, nl = (uint32_t)kv.second.first.size();
        ok = ok && write_all(fd, &id, 4) && write_all(fd, &nl, 4) && write_all(fd, &kv.second.second, 8) && (nl == 0 || write_all(fd, kv.second.first.data(), nl));
    }
    const size_t total = (size_t)n_occ * sizeof(SavedSlot);
    if (ok && total) {
        bool hip_ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2 && hip_ok; ++i)
            hip_ok = hipHostMalloc((void **)&h_buf[i], std::min(total, IX_IO_CHUNK), hipHostMallocDefault) == hipSuccess && hipEventCreate(&ev[i]) == hipSuccess;
        if (!hip_ok) return fail(MQ_EHIP, "mq_index_save: transfer buffers");
        // chunk i+1 crosses PCIe while chunk i goes to the file
        const size_t n_chunks = (total + IX_IO_CHUNK - 1) / IX_IO_CHUNK;
        auto issue = [&](size_t c) {
            const size_t o = c * IX_IO_CHUNK, n = std::min(IX_IO_CHUNK, total - o);
            return hipMemcpyAsync(h_buf[c & 1], (const uint8_t *)d_pack + o, n, hipMemcpyDeviceToHost, st) == hipSuccess &&
                   hipEventRecord(ev[c & 1], st) == hipSuccess;
        };
        hip_ok = issue(0);
        for (size_t c = 0; c < n_chunks && ok && hip_ok; ++c) {
            if (c + 1 < n_chunks) hip_ok = issue(c + 1);
            hip_ok = hip_ok && hipEventSynchronize(ev[c & 1]) == hipSuccess;
            const size_t o = c * IX_IO_CHUNK, n = std::min(IX_IO_CHUNK, total - o);
            ok = hip_ok && write_all(fd, h_buf[c & 1], n);
        }
        if (!hip_ok) return fail(MQ_EHIP, "mq_index_save: device-to-host copy failed");
    }
    const bool closed = ::close(fd) == 0;
    fd = -1;
    cleanup();
    return ok && closed ? MQ_OK : set_err(MQ_EINVAL, std::string("short write: ") + path);
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

mq_index *mq_index_load(const char *path, int device) try {
    if (!path) {
        set_err(MQ_EINVAL, "path is NULL");
        return nullptr;
    }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        set_err(MQ_EINVAL, std::string("cannot open: ") + path);
        return nullptr;
    }
    char magic[8];
    mq_params p;
    uint64_t hdr[6];
    if (!read_all(fd, magic, 8) || memcmp(magic, MQ_INDEX_MAGIC, 8) != 0 || !read_all(fd, &p, sizeof(p)) || !read_all(fd, hdr, sizeof(hdr)) ||
        hdr[0] != sizeof(SavedSlot) || hdr[1] < 2 || (hdr[1] & (hdr[1] - 1)) != 0 || hdr[1] > (1ull << 40) ||
        hdr[3] >= hdr[1] /* a table without an empty slot would make a miss walk forever */ || hdr[4] > hdr[3] || hdr[5] > MQ_MAX_REF_ID) {
        ::close(fd);
        set_err(MQ_EINVAL, std::string("not a mapquik HIP index (or another layout version): ") + path);
        return nullptr;
    }
    mq_index *idx = mq_index_new(&p, device);
    if (!idx) {
        ::close(fd);
        return nullptr;
    }
    bool ok = true;
    for (uint64_t i = 0; ok && i < hdr[5]; ++i) {
        uint32_t id = 0, nl = 0;
        uint64_t len = 0;
        ok = read_all(fd, &id, 4) && read_all(fd, &nl, 4) && read_all(fd, &len, 8) && nl < (1u << 20) && id < MQ_MAX_REF_ID;
        std::string name(nl, '\0');
        ok = ok && (nl == 0 || read_all(fd, &name[0], nl));
        if (ok) idx->refs[id] = std::make_pair(name, len);
    }
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    if (ok && alloc_table(idx, hdr[1]) != MQ_OK) ok = false;
    // file -> page-locked buffer -> device -> scatter kernel, by a few threads at once (each its own buffers and stream; the
    // kernels of different chunks insert into the same table with atomics): the file read, not the copy, is what takes time
    const size_t total = (size_t)hdr[3] * sizeof(SavedSlot);
    const off_t slots_at = ::lseek(fd, 0, SEEK_CUR);
    uint32_t *d_flags = nullptr;
    const char *why = "truncated or unreadable index file: ";
    if (ok && total) {
        ok = slots_at >= 0 && hipMalloc((void **)&d_flags, 4) == hipSuccess && hipMemset(d_flags, 0, 4) == hipSuccess &&
             hipDeviceSynchronize() == hipSuccess;  // the table's memset (null stream) is done before other streams write to it
        const size_t n_chunks = (total + IX_IO_CHUNK - 1) / IX_IO_CHUNK;
        const int n_thr = (int)std::min<size_t>(8, n_chunks);
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        auto work = [&]() {
            uint8_t *h = nullptr, *d = nullptr;
            hipStream_t st = nullptr;
            const size_t cb = std::min(total, IX_IO_CHUNK);
            bool good = hipSetDevice(device) == hipSuccess && (h = (uint8_t *)mq_host_alloc(cb)) != nullptr && hipMalloc((void **)&d, cb) == hipSuccess &&
                        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
            while (good) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks) break;
                const size_t o = c * IX_IO_CHUNK, n = std::min(IX_IO_CHUNK, total - o);
                size_t got = 0;
                while (got < n) {
                    const ssize_t r = ::pread(fd, h + got, n - got, slots_at + (off_t)(o + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got != n) { good = false; break; }
                good = hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st) == hipSuccess;
                const uint64_t ns = n / sizeof(SavedSlot);
                hipLaunchKernelGGL(unpack_slots_kernel, dim3((uint32_t)std::min<uint64_t>((ns + 255) / 256, 1u << 16)), dim3(256), 0, st,
                                   (const SavedSlot *)d, ns, idx->table, hdr[1] - 1, max_id, d_flags);
                good = good && hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            }
            if (!good) bad.store(1);
            if (st) hipStreamDestroy(st);
            hipFree(d);
            mq_host_free(h);
        };
        if (ok) {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; ++t) th.emplace_back(work);
            for (auto &t : th) t.join();
            ok = bad.load() == 0;
        }
        uint32_t flags = 1;
        ok = ok && hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok && flags) {
            ok = false;
            why = "corrupt index file (an entry names a reference the file does not have, or a malformed slot): ";
        }
        if (ok) ok = ::lseek(fd, slots_at + (off_t)total, SEEK_SET) >= 0;
    }
    uint8_t extra = 0;
    if (ok && ::read(fd, &extra, 1) != 0) {
        ok = false;
        why = "corrupt index file (bytes after the last slot): ";
    }
    ::close(fd);
    // what the file says about its table must be what the rebuilt table holds
    if (ok) {
        unsigned long long *d_acc = nullptr, acc[3] = {0, 0, 0};
        ok = hipMalloc((void **)&d_acc, 24) == hipSuccess && hipMemset(d_acc, 0, 24) == hipSuccess;
        if (ok) {
            const uint64_t nb1 = hdr[1] / 2 + 1;
            hipLaunchKernelGGL(count_kernel, dim3((uint32_t)std::min<uint64_t>((2 * nb1 + 255) / 256, 1u << 16)), dim3(256), 0, 0, idx->table, nb1, d_acc);
            ok = hipGetLastError() == hipSuccess && hipMemcpy(acc, d_acc, 24, hipMemcpyDeviceToHost) == hipSuccess;
        }
        hipFree(d_acc);
        if (ok && (acc[1] != hdr[3] || acc[0] != hdr[4] || (acc[2] != 0 && acc[2] - 1 > max_id))) {
            ok = false;
            why = "corrupt index file (key counts or reference ids disagree with its header): ";
        }
    }
    if (ok) {
        std::vector<uint64_t> lens((size_t)max_id + 1, 0);
        for (auto &kv : idx->refs) lens[kv.first] = kv.second.second;
        ok = hipMalloc((void **)&idx->d_ref_lens, lens.size() * sizeof(uint64_t)) == hipSuccess &&
             hipMemcpy(idx->d_ref_lens, lens.data(), lens.size() * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    hipFree(d_flags);
    if (!ok) {
        mq_index_free(idx);
        set_err(MQ_EINVAL, std::string(why) + path);
        return nullptr;
    }
    idx->n_kmm_total = hdr[2];
    idx->n_keys = hdr[3];
    idx->n_unique = hdr[4];
    idx->finalized = true;
    return idx;
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

// A replica of a finalized index on another device: the table travels device to device (xGMI between the GPUs of a node)
// instead of being rebuilt from the reference on every GPU.
mq_index *mq_index_clone(const mq_index *src, int device) try {
    if (!src) {
        set_err(MQ_EINVAL, "src is NULL");
        return nullptr;
    }
    if (!src->finalized) {
        set_err(MQ_ESTATE, "index not finalized");
        return nullptr;
    }
    mq_index *idx = mq_index_new(&src->params, device);
    if (!idx) return nullptr;
    idx->refs = src->refs;
    idx->n_kmm_total = src->n_kmm_total;
    idx->n_keys = src->n_keys;
    idx->n_unique = src->n_unique;
    bool ok = alloc_table(idx, src->nslots) == MQ_OK;
    if (ok) ok = hipMemcpyPeer(idx->table, device, src->table, src->device, table_bytes_of(src->nslots)) == hipSuccess;
    uint32_t max_id = 0;
    for (auto &kv : idx->refs) max_id = std::max(max_id, kv.first);
    const size_t nl = (size_t)max_id + 1;
    if (ok) ok = hipSetDevice(device) == hipSuccess && hipMalloc((void **)&idx->d_ref_lens, nl * sizeof(uint64_t)) == hipSuccess;
    if (ok) ok = hipMemcpyPeer(idx->d_ref_lens, device, src->d_ref_lens, src->device, nl * sizeof(uint64_t)) == hipSuccess;
    if (ok) ok = hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        mq_index_free(idx);
        set_err(MQ_EHIP, "mq_index_clone: device-to-device copy failed");
        return nullptr;
    }
    idx->finalized = true;
    return idx;
} catch (const std::bad_alloc &) {
    set_err(MQ_ENOMEM, "out of host memory");
    return nullptr;
} catch (const std::exception &e) {
    set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
    return nullptr;
}

// The parameters an index was built with (what a loaded file says: k, l, density, use_hpc and the seeding variant decide its keys).
int mq_index_get_params(const mq_index *idx, mq_params *out) try {
    if (!idx || !out) return set_err(MQ_EINVAL, "bad arguments");
    *out = idx->params;
    return MQ_OK;
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

// The parameters that act at mapping time only (Params.c / .s in Chain::get_match src/chain.rs:147-169, .g in the gap tests
// src/chain.rs:132-142, and the feeder's case folding): a loaded index takes the command line's.  Launches queued before the call keep
// the old values.
int mq_index_set_map_params(mq_index *idx, uint32_t c, uint32_t s, uint32_t g, int fold_case) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->params.c = c;
    idx->params.s = s;
    idx->params.g = g;
    idx->params.flags = (idx->params.flags & ~MQ_FLAG_FOLD_CASE) | (fold_case ? MQ_FLAG_FOLD_CASE : 0u);
    idx->dp.c = c;
    idx->dp.s = s;
    idx->dp.g = g;
    idx->dp.fold = fold_case ? 1u : 0u;
    return MQ_OK;
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_index_ref_info(const mq_index *idx, uint32_t ref_id, const char **name, uint64_t *len) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    auto it = idx->refs.find(ref_id);
    if (it == idx->refs.end()) return set_err(MQ_EINVAL, "unknown ref_id");
    if (name) *name = it->second.first.c_str();
    if (len) *len = it->second.second;
    return MQ_OK;
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

int mq_map_reserve(mq_index *idx, uint32_t n_reads, uint64_t total_bases) try {
    if (!idx) return set_err(MQ_EINVAL, "idx is NULL");
    std::lock_guard<std::mutex> lk(idx->mu);
    int rc = use_device(idx);
    if (rc) return rc;
    return ctx_ensure(idx->def_ctx, n_reads, total_bases, list_f16(idx));
} catch (const std::bad_alloc &) {
    return set_err(MQ_ENOMEM, "out of host memory");
} catch (const std::exception &e) {
    return set_err(MQ_EINVAL, std::string("unexpected exception: ") + e.what());
}

}  // extern "C"

// mq_fastx.hpp -- FASTA records found on the device (closures.rs:100-123 hands every record's id and sequence to the mapper; here
// the host never looks at a base: a chunk of raw file bytes that holds whole records goes to the device as it is, these kernels find
// the line ends, and map_kernel takes the sequence spans from device memory).
//   count_newlines_kernel   '\n' per 16-KB tile (one wave per tile, 16-byte lane loads)
//   scan_tiles_kernel       exclusive scan of the tile counts (one workgroup); appends the virtual line end of a last line without '\n'
//   list_newlines_kernel    every '\n' position, in order, at its place
//   fasta_spans_kernel      record r = lines 2r (header, starts with '>') and 2r + 1 (sequence): start / length of the sequence line (a
//                           '\r' in front of the '\n' is cut); anything else -- a sequence over several lines, blank lines, a line that
//                           does not start with '>' where a header must be -- sets the IRREGULAR flag and the host parses the chunk itself
// Integer / byte work, HBM-stream bound: a 32-MB chunk is scanned twice (64 MB of reads) in ~30 us.
#pragma once
#include "mq_device.hpp"

namespace mq {

constexpr uint32_t FX_TILE = 16384;  // bytes per wave and pass: 16 iterations of 64 lanes x 16 bytes
constexpr uint32_t FX_IRREGULAR = 1u;

// per byte of a dword: 0x80 where the byte equals '\n' (exact: no borrow between bytes)
__device__ __forceinline__ uint32_t nl_mask32(uint32_t w) {
    const uint32_t x = w ^ 0x0A0A0A0Au;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}
// bit b of the result: byte b of the 16 bytes is '\n' and its position p0 + b lies in [begin, end)
__device__ __forceinline__ uint32_t nl_bits16(const uint4 v, uint32_t p0, uint32_t begin, uint32_t end) {
    auto pack = [](uint32_t m) { return ((m >> 7) & 1u) | ((m >> 14) & 2u) | ((m >> 21) & 4u) | ((m >> 28) & 8u); };
    uint32_t bits = pack(nl_mask32(v.x)) | (pack(nl_mask32(v.y)) << 4) | (pack(nl_mask32(v.z)) << 8) | (pack(nl_mask32(v.w)) << 12);
    if (p0 < begin) bits &= begin - p0 >= 16u ? 0u : (0xFFFFu << (begin - p0));
    if (p0 + 16u > end) bits &= p0 >= end ? 0u : (0xFFFFu >> (p0 + 16u - end));
    return bits;
}

// buf: 16-byte aligned, readable up to the next multiple of 16 behind `end`
__global__ __launch_bounds__(256) void count_newlines_kernel(const uint8_t *__restrict__ buf, uint32_t begin, uint32_t end, uint32_t n_tiles,
                                                             uint32_t *__restrict__ tile_counts) {
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        uint32_t c = 0;
#pragma unroll 4
        for (uint32_t it = 0; it < FX_TILE / 1024u; ++it) {
            const uint32_t p0 = t * FX_TILE + it * 1024u + lane * 16u;
            if (p0 < end) c += (uint32_t)__popc(nl_bits16(*reinterpret_cast<const uint4 *>(buf + p0), p0, begin, end));
        }
        c = wave_sum_u32(c);
        if (lane == 0) tile_counts[t] = c;
    }
}

// info: [0] lines (incl. the virtual end of a last line without '\n'), [1] records, [2] flags, [3] spare
__global__ __launch_bounds__(1024) void scan_tiles_kernel(const uint8_t *__restrict__ buf, uint32_t begin, uint32_t end, const uint32_t *__restrict__ tile_counts,
                                                          uint32_t n_tiles, uint32_t *__restrict__ tile_off, uint32_t *__restrict__ nl_pos, uint32_t nl_cap,
                                                          uint32_t *__restrict__ info, uint32_t lines_per_record) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (n_tiles + 1023u) / 1024u;
    const uint32_t lo = t * per < n_tiles ? t * per : n_tiles, hi = lo + per < n_tiles ? lo + per : n_tiles;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += tile_counts[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        tile_off[i] = run;
        run += tile_counts[i];
    }
    if (t == 1023u) {
        uint32_t lines = part[1023];
        if (end > begin && buf[end - 1] != '\n') {  // the file's last line has no '\n': it ends where the data ends
            if (lines < nl_cap) nl_pos[lines] = end;
            lines++;
        }
        // more line ends than nl_pos holds (records shorter than ~32 bytes: primers, barcodes, junk): the piece is IRREGULAR and NO record
        // is reported -- fasta_spans_kernel and the caller index nl_pos[0, lines) and must never see a count beyond its capacity
        const bool over = lines > nl_cap;
        info[0] = over ? 0u : lines;
        info[1] = over ? 0u : lines / lines_per_record;  // FASTA: header + sequence line; FASTQ: header, sequence, '+', quality
        info[2] = (lines % lines_per_record) || over ? FX_IRREGULAR : 0u;
        info[3] = 0;
    }
}

__global__ __launch_bounds__(256) void list_newlines_kernel(const uint8_t *__restrict__ buf, uint32_t begin, uint32_t end, uint32_t n_tiles,
                                                            const uint32_t *__restrict__ tile_off, uint32_t *__restrict__ nl_pos, uint32_t nl_cap) {
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        uint32_t at = tile_off[t];
        for (uint32_t it = 0; it < FX_TILE / 1024u; ++it) {
            const uint32_t p0 = t * FX_TILE + it * 1024u + lane * 16u;
            uint32_t bits = 0;
            if (p0 < end) bits = nl_bits16(*reinterpret_cast<const uint4 *>(buf + p0), p0, begin, end);
            if (__ballot(bits != 0u) == 0) continue;  // line ends are rare (one per read)
            const uint32_t mine = (uint32_t)__popc(bits);
            const uint32_t incl = wave_incl_scan_u32(mine);
            uint32_t o = at + incl - mine;
            while (bits) {
                const uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
                if (o < nl_cap) nl_pos[o] = p0 + b;
                ++o;
                bits &= bits - 1u;
            }
            at += rdlane(incl, 63);
        }
    }
}

// starts[r] / lens[r]: the sequence line of record r; info[2] |= FX_IRREGULAR when the chunk is not "header line, sequence line" all through
__global__ __launch_bounds__(256) void fasta_spans_kernel(const uint8_t *__restrict__ buf, uint32_t begin, uint32_t end, const uint32_t *__restrict__ nl_pos,
                                                          uint32_t *__restrict__ info, unsigned long long *__restrict__ starts, uint32_t *__restrict__ lens,
                                                          uint32_t span_cap) {
    if (info[2] & FX_IRREGULAR) return;  // decided by the scan already (odd line count / more lines than nl_pos holds): nothing to look at
    const uint32_t n_rec = info[1];
    bool bad = false;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += gridDim.x * blockDim.x) {
        const uint32_t hs = r == 0 ? begin : nl_pos[2u * r - 1u] + 1u;  // header line [hs, he), sequence line [he + 1, se)
        const uint32_t he = nl_pos[2u * r], se = nl_pos[2u * r + 1u];
        uint32_t ss = he + 1u, e = se;
        if (hs >= he || buf[hs] != '>') bad = true;            // a blank line, or a sequence that goes on over several lines
        if (ss < e && buf[ss] == '>') bad = true;              // a header without a sequence line
        if (e > ss && buf[e - 1u] == '\r') --e;                // CR-LF
        if (r < span_cap) {
            starts[r] = ss;
            lens[r] = e - ss;
        } else {
            bad = true;
        }
    }
    if (__ballot(bad) && lane_id() == 0) atomicOr(&info[2], FX_IRREGULAR);
}

// FASTQ (the reference's default format when the name is not .fa / .fasta / .fna, src/main.rs:196-205; its headline real-data run is an
// uncompressed FASTQ, experiments/table1.sh:50): record r = lines 4r .. 4r + 3 = "@id ...", sequence, "+...", quality.  Checked per record,
// as the host's validator does (fastx_records.hpp fastq_record_at): '@' opens the header line, '+' the third line, and the quality line
// is as long as the sequence line ('\r' cut from both).  Anything else -- sequences or qualities over several lines, blank lines, a
// record cut short -- sets FX_IRREGULAR and the host parses the piece.  The quality bytes cross the link (2 file bytes per base) but no
// host thread touches them, and no kernel reads more of them than the byte in front of their line end.
__global__ __launch_bounds__(256) void fastq_spans_kernel(const uint8_t *__restrict__ buf, uint32_t begin, uint32_t end, const uint32_t *__restrict__ nl_pos,
                                                          uint32_t *__restrict__ info, unsigned long long *__restrict__ starts, uint32_t *__restrict__ lens,
                                                          uint32_t span_cap) {
    if (info[2] & FX_IRREGULAR) return;
    const uint32_t n_rec = info[1];
    bool bad = false;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rec; r += gridDim.x * blockDim.x) {
        const uint32_t hs = r == 0 ? begin : nl_pos[4u * r - 1u] + 1u;
        const uint32_t he = nl_pos[4u * r], se = nl_pos[4u * r + 1u], pe = nl_pos[4u * r + 2u], qe = nl_pos[4u * r + 3u];
        const uint32_t ss = he + 1u, ps = se + 1u, qs = pe + 1u;
        uint32_t e = se, q = qe;
        if (hs >= he || buf[hs] != '@') bad = true;       // a blank line, or not a header where one must be
        if (ps >= pe || buf[ps] != '+') bad = true;       // the third line is the separator
        if (e > ss && buf[e - 1u] == '\r') --e;           // CR-LF
        if (q > qs && q <= end && buf[q - 1u] == '\r') --q;
        if (e - ss != q - qs) bad = true;                 // one quality per base
        if (r < span_cap) {
            starts[r] = ss;
            lens[r] = e - ss;
        } else {
            bad = true;
        }
    }
    if (__ballot(bad) && lane_id() == 0) atomicOr(&info[2], FX_IRREGULAR);
}

}  // namespace mq

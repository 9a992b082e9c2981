/*
 * rccl_reducer.h — the multi-GPU exchange of a tile-sharded render, in C++ over RCCL (no Python, no torch).
 *
 * One process per GPU (`ppg_render --rank R --world N --nccl-id FILE ...`); every rank holds full replicas of BVH and SD-tree and renders
 * the 32x32 tiles t with t % world == rank (ppg_set_shard).  What is exchanged:
 *
 *   per iteration, before the variance estimate   image + squared image + weights of the iteration: disjoint supports, float sums are exact
 *   per iteration, before buildSDTree             the building tree's fixed-point leaf sums and per-D-tree statistical weights (uint64 as int64:
 *                                                 integer sums are exact and order independent → refine / reset / build stay identical on all ranks)
 *   per round of the sampling-fraction optimiser  its records (include/ppg.h): every rank applies the union in key order
 *   at the end                                    the film (not with inverse-variance combination: the retained iteration images were reduced)
 *
 * Each exchange is ONE collective: the arrays of an exchange are packed into a staging buffer on the device (they are separate allocations of
 * the context), all-reduced / all-gathered in place, and unpacked — xGMI rings are latency bound for these sizes (a few MB to ~100 MB), so
 * fewer, larger calls beat one call per array.  The communicator is bootstrapped through a file holding the ncclUniqueId (rank 0 writes it).
 */
#ifndef PPG_RCCL_REDUCER_H
#define PPG_RCCL_REDUCER_H

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ppg.h"
#include "guided_path_hip.h"

namespace ppg {

class RcclReducer : public Reducer {
public:
    RcclReducer(int rank, int world, int device, const std::string &idFile) : m_rank(rank), m_world(world) {
        hip(hipSetDevice(device), "hipSetDevice");
        ncclUniqueId id;
        if (rank == 0) {
            nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
            std::ofstream f(idFile + ".tmp", std::ios::binary);
            f.write((const char *)&id, sizeof id);
            f.close();
            if (std::rename((idFile + ".tmp").c_str(), idFile.c_str()) != 0) throw std::runtime_error("cannot write " + idFile);
        } else {
            for (int tries = 0;; ++tries) {
                std::ifstream f(idFile, std::ios::binary);
                if (f && f.read((char *)&id, sizeof id)) break;
                if (tries > 6000) throw std::runtime_error("timed out waiting for " + idFile);
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        nccl(ncclCommInitRank(&m_comm, world, id, rank), "ncclCommInitRank");
        hip(hipStreamCreate(&m_stream), "hipStreamCreate");
    }
    ~RcclReducer() {
        if (m_stage) (void)hipFree(m_stage);
        if (m_counts) (void)hipFree(m_counts);
        if (m_comm) (void)ncclCommDestroy(m_comm);
        if (m_stream) (void)hipStreamDestroy(m_stream);
    }
    int rank() const override { return m_rank; }
    int world() const override { return m_world; }
    size_t collectives() const { return m_collectives; }
    size_t bytes() const { return m_bytes; }

    void reduceImages(ppg_ctx *ctx, int width, int height) override {
        void *img, *sq, *w;
        check(ctx, ppg_image_buffers(ctx, &img, &sq, &w), "ppg_image_buffers");
        const size_t n = (size_t)width * height;
        Piece p[3] = {{img, 3 * n * 4}, {sq, 3 * n * 4}, {w, n * 4}};
        allReduce(p, 3, ncclFloat);
    }
    void reduceSDTree(ppg_ctx *ctx) override {
        void *sums, *weights;
        uint64_t ns, nw;
        check(ctx, ppg_sdtree_stat_buffers(ctx, &sums, &ns, &weights, &nw), "ppg_sdtree_stat_buffers");
        Piece p[2] = {{sums, (size_t)ns * 8}, {weights, (size_t)nw * 8}};
        allReduce(p, 2, ncclInt64);
    }
    void reduceFilm(ppg_ctx *ctx, int width, int height) override {
        void *rgb, *w;
        check(ctx, ppg_film_buffers(ctx, &rgb, &w), "ppg_film_buffers");
        const size_t n = (size_t)width * height;
        Piece p[2] = {{rgb, 3 * n * 4}, {w, n * 4}};
        allReduce(p, 2, ncclFloat);
    }
    // every rank applies the records of all ranks: counts first, then one padded all-gather, then the pieces packed back to back
    void reduceAdamRecords(ppg_ctx *ctx) override {
        void *recs;
        uint64_t n;
        check(ctx, ppg_adam_records(ctx, &recs, &n), "ppg_adam_records");
        if (!m_counts) hip(hipMalloc(&m_counts, (size_t)m_world * sizeof(unsigned long long)), "hipMalloc");
        std::vector<unsigned long long> counts((size_t)m_world, 0ull);
        unsigned long long mine = n;
        hip(hipMemcpyAsync((unsigned long long *)m_counts + m_rank, &mine, 8, hipMemcpyHostToDevice, m_stream), "hipMemcpyAsync");
        nccl(ncclAllGather((unsigned long long *)m_counts + m_rank, m_counts, 1, ncclUint64, m_comm, m_stream), "ncclAllGather(counts)");
        hip(hipMemcpyAsync(counts.data(), m_counts, (size_t)m_world * 8, hipMemcpyDeviceToHost, m_stream), "hipMemcpyAsync");
        hip(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        ++m_collectives;
        unsigned long long most = 0, total = 0;
        for (unsigned long long c : counts) { most = std::max(most, c); total += c; }
        if (most == 0) return;
        const size_t rec = sizeof(ppg_adam_record), slot = (size_t)most * rec;
        reserve(slot * (size_t)m_world + (size_t)total * rec);
        char *gather = (char *)m_stage, *packed = gather + slot * (size_t)m_world;
        if (n) hip(hipMemcpyAsync(gather + slot * (size_t)m_rank, recs, (size_t)n * rec, hipMemcpyDeviceToDevice, m_stream), "hipMemcpyAsync");
        nccl(ncclAllGather(gather + slot * (size_t)m_rank, gather, slot, ncclChar, m_comm, m_stream), "ncclAllGather(records)");
        size_t off = 0;
        for (int r = 0; r < m_world; ++r) {
            if (counts[r]) hip(hipMemcpyAsync(packed + off, gather + slot * (size_t)r, (size_t)counts[r] * rec, hipMemcpyDeviceToDevice, m_stream), "hipMemcpyAsync");
            off += (size_t)counts[r] * rec;
        }
        hip(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        ++m_collectives; m_bytes += slot * (size_t)m_world;
        check(ctx, ppg_adam_records_replace(ctx, packed, total), "ppg_adam_records_replace");
    }

private:
    struct Piece { void *dev; size_t bytes; };
    void reserve(size_t bytes) {
        if (bytes <= m_stageBytes) return;
        if (m_stage) (void)hipFree(m_stage);
        m_stageBytes = bytes + bytes / 4;
        hip(hipMalloc(&m_stage, m_stageBytes), "hipMalloc");
    }
    // pack → one in-place all-reduce → unpack
    void allReduce(const Piece *p, int count, ncclDataType_t type) {
        size_t total = 0;
        for (int i = 0; i < count; ++i) total += p[i].bytes;
        if (total == 0) return;
        reserve(total);
        size_t off = 0;
        for (int i = 0; i < count; ++i) { if (p[i].bytes) hip(hipMemcpyAsync((char *)m_stage + off, p[i].dev, p[i].bytes, hipMemcpyDeviceToDevice, m_stream), "hipMemcpyAsync"); off += p[i].bytes; }
        const size_t elem = type == ncclFloat ? 4 : 8;
        nccl(ncclAllReduce(m_stage, m_stage, total / elem, type, ncclSum, m_comm, m_stream), "ncclAllReduce");
        off = 0;
        for (int i = 0; i < count; ++i) { if (p[i].bytes) hip(hipMemcpyAsync(p[i].dev, (char *)m_stage + off, p[i].bytes, hipMemcpyDeviceToDevice, m_stream), "hipMemcpyAsync"); off += p[i].bytes; }
        hip(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        ++m_collectives; m_bytes += total;
    }
    static void hip(hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
    static void nccl(ncclResult_t r, const char *what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
    static void check(ppg_ctx *ctx, int rc, const char *what) { if (rc != PPG_OK) throw std::runtime_error(std::string(what) + ": " + ppg_last_error(ctx)); }

    int m_rank, m_world;
    ncclComm_t m_comm = nullptr;
    hipStream_t m_stream = nullptr;
    void *m_stage = nullptr, *m_counts = nullptr;
    size_t m_stageBytes = 0, m_collectives = 0, m_bytes = 0;
};

}  // namespace ppg
#endif

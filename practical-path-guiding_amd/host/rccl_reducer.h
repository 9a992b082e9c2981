/*
 * rccl_reducer.h — the multi-GPU exchange of a tile-sharded render, in C++ over RCCL (no Python, no torch).
 *
 * One process per GPU (`ppg_render --rank R --world N --nccl-id FILE ...`); every rank holds full replicas of BVH and SD-tree and renders
 * the 32x32 tiles t with t % world == rank (ppg_set_shard).  What is exchanged:
 *
 *   per iteration, before the variance estimate   image + squared image + weights of the iteration: disjoint supports, float sums are exact
 *   per iteration, before buildSDTree             the building tree's fixed-point leaf sums and per-D-tree statistical weights (uint64 as int64:
 *                                                 integer sums are exact and order independent → refine / reset / build stay identical on all ranks)
 *   per round of the sampling-fraction optimiser  ONE OWNER PER D-TREE (include/ppg.h "Sharded optimiser"; the reference serialises Adam per D-tree, GP:719-737):
 *                                                 phase 0 — every record goes to the owner of its D-tree (grouped ncclSend / ncclRecv = all-to-all-v, a rank
 *                                                 receives 1 / world of the records), which sorts and applies them in key order; phase 1 — the owners'
 *                                                 optimiser state (24 B per S-tree node) is all-gathered in place.  Same key order at the owner ⇒ same bits
 *   at the end                                    the film (not with inverse-variance combination: the retained iteration images were reduced)
 *
 * Each exchange is ONE collective: the arrays of an exchange are packed into a staging buffer on the device (they are separate allocations of
 * the context), all-reduced / all-gathered in place, and unpacked — xGMI rings are latency bound for these sizes (a few MB to ~100 MB), so
 * fewer, larger calls beat one call per array.  The communicator is bootstrapped through a file holding the ncclUniqueId: rank 0 removes
 * what a previous run may have left, writes {run tag, id} under a temporary name and renames it; the other ranks accept the file only if
 * it carries THEIR run tag (`--run-tag` / PPG_RUN_TAG: any string the launcher gives to all ranks of one run), and rank 0 removes the file again
 * once the communicator exists.
 * Every exchange also carries one status word per rank (0 = fine): a rank that was cancelled or failed says so in the next exchange, all
 * ranks see the same sum and leave together instead of waiting for each other in a collective that will never complete.
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
    RcclReducer(int rank, int world, int device, const std::string &idFile, const std::string &runTag = std::string()) : m_rank(rank), m_world(world) {
        hip(hipSetDevice(device), "hipSetDevice");
        ncclUniqueId id;
        char tag[64];
        memset(tag, 0, sizeof tag);
        strncpy(tag, runTag.c_str(), sizeof tag - 1);
        if (rank == 0) {
            (void)std::remove(idFile.c_str());  // whatever an earlier run left behind
            nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
            std::ofstream f(idFile + ".tmp", std::ios::binary);
            f.write(tag, sizeof tag);
            f.write((const char *)&id, sizeof id);
            f.close();
            if (!f || std::rename((idFile + ".tmp").c_str(), idFile.c_str()) != 0) throw std::runtime_error("cannot write " + idFile);
        } else {
            for (int tries = 0;; ++tries) {
                std::ifstream f(idFile, std::ios::binary);
                char got[64];
                if (f && f.read(got, sizeof got) && f.read((char *)&id, sizeof id) && memcmp(got, tag, sizeof tag) == 0) break;  // (a stale file carries another tag)
                if (tries > 6000) throw std::runtime_error("timed out waiting for " + idFile + " with this run's tag");
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        nccl(ncclCommInitRank(&m_comm, world, id, rank), "ncclCommInitRank");
        if (rank == 0) (void)std::remove(idFile.c_str());  // every rank has read it: ncclCommInitRank returns when all have joined
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
    void reduceFinalPartials(ppg_ctx *ctx, void *dev, uint64_t nFloats) override {
        // film head + every group's partial image: each slot is non-zero on one rank, the float sums are exact (include/ppg.h)
        // (dev == nullptr: this rank has nothing to give — it failed or was cancelled — but joins the collective, zeros and its status word)
        // ONE piece: reduced in place (no staging copy of what can be gigabytes), the status word in a small collective of its own behind it
        if (dev) {
            nccl(ncclAllReduce(dev, dev, (size_t)nFloats, ncclFloat, ncclSum, m_comm, m_stream), "ncclAllReduce(partials)");
            ++m_collectives; m_bytes += (size_t)nFloats * 4;
            allReduce(nullptr, 0, ncclFloat);
        } else {
            Piece p[1] = {{nullptr, (size_t)nFloats * 4}};  // (zeros from the staging block)
            allReduce(p, 1, ncclFloat, true);
        }
        check(ctx, ppg_final_partials_commit(ctx), "ppg_final_partials_commit");
    }
    // a rank that was cancelled or failed announces it in the next exchange (status word, see the header comment)
    void setLocalStatus(int status) override { m_status = status; }
    void beginRender() override { m_status = 0; }
    double broadcast(double value) override {
        if (!m_counts) hip(hipMalloc(&m_counts, (size_t)m_world * ((size_t)m_world + 1) * 8), "hipMalloc");
        hip(hipMemcpyAsync(m_counts, &value, 8, hipMemcpyHostToDevice, m_stream), "hipMemcpyAsync");
        nccl(ncclBroadcast(m_counts, m_counts, 1, ncclDouble, 0, m_comm, m_stream), "ncclBroadcast");
        hip(hipMemcpyAsync(&value, m_counts, 8, hipMemcpyDeviceToHost, m_stream), "hipMemcpyAsync");
        hip(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        ++m_collectives;
        return value;
    }

    int stopDecision(int localStop) override {
        // one all-reduce (sum) of {rank 0's decision, every rank's status word}
        if (!m_counts) hip(hipMalloc(&m_counts, (size_t)m_world * ((size_t)m_world + 1) * 8), "hipMalloc");
        long long v[2] = {m_rank == 0 ? (long long)(localStop != 0) : 0ll, (long long)m_status};
        hip(hipMemcpyAsync(m_counts, v, 16, hipMemcpyHostToDevice, m_stream), "hipMemcpyAsync");
        nccl(ncclAllReduce(m_counts, m_counts, 2, ncclInt64, ncclSum, m_comm, m_stream), "ncclAllReduce(stop)");
        hip(hipMemcpyAsync(v, m_counts, 16, hipMemcpyDeviceToHost, m_stream), "hipMemcpyAsync");
        hip(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        ++m_collectives;
        return (v[0] != 0 || v[1] != 0) ? 1 : 0;
    }

    // round hook of the sampling-fraction optimiser, called twice per round (include/ppg.h "Sharded optimiser")
    void reduceAdamRecords(ppg_ctx *ctx) override {
        int32_t phase = 0;
        check(ctx, ppg_hook_phase(ctx, &phase), "ppg_hook_phase");
        const size_t rec = sizeof(ppg_adam_record), W = (size_t)m_world;
        if (phase == 0) {
            // who gets how many of my records: one all-gather of a (world + 1)-vector per rank — counts per owner + my status
            // (the exchanges of the hook run on the CONTEXT's stream: what the library enqueues next — sort, apply, commit — follows them
            // in stream order, no host synchronisation in between except where the host needs the counts)
            void *cs = nullptr;
            check(ctx, ppg_exchange_stream(ctx, &cs), "ppg_exchange_stream");
            hipStream_t st = (hipStream_t)cs;
            void *recs = nullptr;
            std::vector<uint64_t> send(W, 0);
            // a rank that cannot take part (cancelled, or its records cannot be produced) still joins the count exchange — with zero records
            // and its status word set —, so that all ranks see the same sum and leave the round together
            std::string localError;
            if (m_status == 0 && ppg_adam_records_by_owner(ctx, m_world, &recs, send.data()) != PPG_OK) {
                localError = std::string("ppg_adam_records_by_owner: ") + ppg_last_error(ctx);
                m_status = 1;
            }
            if (m_status) std::fill(send.begin(), send.end(), 0);
            const size_t row = W + 1;
            if (!m_counts) hip(hipMalloc(&m_counts, W * row * 8), "hipMalloc");
            std::vector<unsigned long long> mine(row, 0ull), all(W * row, 0ull);
            for (size_t r = 0; r < W; ++r) mine[r] = send[r];
            mine[W] = (unsigned long long)m_status;
            unsigned long long *dc = (unsigned long long *)m_counts;
            hip(hipMemcpyAsync(dc + (size_t)m_rank * row, mine.data(), row * 8, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
            nccl(ncclAllGather(dc + (size_t)m_rank * row, dc, row, ncclUint64, m_comm, st), "ncclAllGather(counts)");
            hip(hipMemcpyAsync(all.data(), dc, W * row * 8, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
            hip(hipStreamSynchronize(st), "hipStreamSynchronize");  // (the host sizes the messages by the counts)
            ++m_collectives;
            unsigned long long bad = 0, total = 0;
            std::vector<unsigned long long> recv(W, 0ull);
            for (size_t r = 0; r < W; ++r) { bad += all[r * row + W]; recv[r] = all[r * row + (size_t)m_rank]; total += recv[r]; }
            if (bad) throw std::runtime_error(localError.empty() ? std::string("render aborted: a rank reported a failure or a cancellation") : localError);
            // all-to-all-v: grouped point-to-point, one message per pair of ranks that has records for each other
            reserve(std::max<size_t>((size_t)total * rec, 64));
            nccl(ncclGroupStart(), "ncclGroupStart");
            size_t soff = 0, roff = 0;
            for (size_t r = 0; r < W; ++r) {
                if (send[r]) nccl(ncclSend((const char *)recs + soff, (size_t)send[r] * rec, ncclChar, (int)r, m_comm, st), "ncclSend");
                if (recv[r]) nccl(ncclRecv((char *)m_stage + roff, (size_t)recv[r] * rec, ncclChar, (int)r, m_comm, st), "ncclRecv");
                soff += (size_t)send[r] * rec; roff += (size_t)recv[r] * rec;
            }
            nccl(ncclGroupEnd(), "ncclGroupEnd");
            ++m_collectives; m_bytes += (size_t)total * rec;
            check(ctx, ppg_adam_records_replace(ctx, m_stage, total), "ppg_adam_records_replace");
        } else {
            void *state;
            uint64_t seg = 0;
            check(ctx, ppg_adam_state(ctx, m_world, &state, &seg), "ppg_adam_state");
            void *cs = nullptr;
            check(ctx, ppg_exchange_stream(ctx, &cs), "ppg_exchange_stream");
            const size_t bytes = (size_t)seg * 24;
            if (bytes) nccl(ncclAllGather((const char *)state + (size_t)m_rank * bytes, state, bytes, ncclChar, m_comm, (hipStream_t)cs), "ncclAllGather(state)");
            ++m_collectives; m_bytes += bytes * W;
            check(ctx, ppg_adam_state_commit(ctx), "ppg_adam_state_commit");
        }
    }

private:
    struct Piece { void *dev; size_t bytes; };
    void reserve(size_t bytes) {
        if (bytes <= m_stageBytes) return;
        (void)hipDeviceSynchronize();  // (work enqueued on the context's stream may still read the old block)
        if (m_stage) (void)hipFree(m_stage);
        m_stageBytes = bytes + bytes / 4;
        hip(hipMalloc(&m_stage, m_stageBytes), "hipMalloc");
    }
    // pack → one in-place all-reduce → unpack
    // statusApart: the status word travels in a second collective (the peers reduce their data in place and the word on its own)
    void allReduce(const Piece *p, int count, ncclDataType_t type, bool statusApart = false) {
        size_t total = 0;
        for (int i = 0; i < count; ++i) total += p[i].bytes;
        const size_t elem = type == ncclFloat ? 4 : 8;
        reserve(total + elem);  // + the status word (summed like the data: > 0 means some rank gave up)
        size_t off = 0;
        for (int i = 0; i < count; ++i) {
            if (p[i].bytes && p[i].dev) hip(hipMemcpyAsync((char *)m_stage + off, p[i].dev, p[i].bytes, hipMemcpyDeviceToDevice, m_stream), "hipMemcpyAsync");
            else if (p[i].bytes) hip(hipMemsetAsync((char *)m_stage + off, 0, p[i].bytes, m_stream), "hipMemsetAsync");
            off += p[i].bytes;
        }
        const float sf = (float)m_status; const long long si = (long long)m_status;
        hip(hipMemcpyAsync((char *)m_stage + total, type == ncclFloat ? (const void *)&sf : (const void *)&si, elem, hipMemcpyHostToDevice, m_stream), "hipMemcpyAsync");
        if (statusApart) {
            nccl(ncclAllReduce(m_stage, m_stage, total / elem, type, ncclSum, m_comm, m_stream), "ncclAllReduce");
            nccl(ncclAllReduce((char *)m_stage + total, (char *)m_stage + total, 1, type, ncclSum, m_comm, m_stream), "ncclAllReduce(status)");
        } else
        nccl(ncclAllReduce(m_stage, m_stage, total / elem + 1, type, ncclSum, m_comm, m_stream), "ncclAllReduce");
        float rf = 0; long long ri = 0;
        hip(hipMemcpyAsync(type == ncclFloat ? (void *)&rf : (void *)&ri, (char *)m_stage + total, elem, hipMemcpyDeviceToHost, m_stream), "hipMemcpyAsync");
        off = 0;
        for (int i = 0; i < count; ++i) { if (p[i].bytes && p[i].dev) hip(hipMemcpyAsync(p[i].dev, (char *)m_stage + off, p[i].bytes, hipMemcpyDeviceToDevice, m_stream), "hipMemcpyAsync"); off += p[i].bytes; }
        hip(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        ++m_collectives; m_bytes += total;
        if (rf != 0 || ri != 0) throw std::runtime_error("render aborted: a rank reported a failure or a cancellation");
    }
    static void hip(hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); }
    static void nccl(ncclResult_t r, const char *what) { if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r)); }
    static void check(ppg_ctx *ctx, int rc, const char *what) { if (rc != PPG_OK) throw std::runtime_error(std::string(what) + ": " + ppg_last_error(ctx)); }

    int m_rank, m_world;
    ncclComm_t m_comm = nullptr;
    hipStream_t m_stream = nullptr;
    void *m_stage = nullptr, *m_counts = nullptr;
    size_t m_stageBytes = 0, m_collectives = 0, m_bytes = 0;
    int m_status = 0;
};

}  // namespace ppg
#endif

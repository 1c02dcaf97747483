// BASELINE config 4 ("examples/HomomRLWR.hs pipeline, 8192-ciphertext batch sharded over 8 x MI355X") from a native host: no Python,
// no torch -- one process, one host thread per GPU, RCCL through include/alchemy_rccl.h.
//
//   ringround_multi [--gpus N] [--batch B] [--lanes S] [--passes K] [--gather G] [--fixture PATH]
//   ringround_multi --world N --rank R --id-file PATH [--device D] [...]      one PROCESS per GPU (the launch model of torchrun /
//       mpirun, natively): start N copies, one per rank; rank 0 writes RCCL's 128-byte id to PATH (alch_comm_unique_id), the others
//       wait for the file and join with alch_comm_init_rank on device D (default: R modulo the visible devices).  Steps 1-5 are the
//       same with one local rank per process; every process prints its own JSON line and exit status.
//
//   1. every rank r (thread r, device r) builds the pipeline of alchemy_amd/host/ringround.hpp for its shard of B ciphertexts
//      (default 1024: 8 ranks x 1024 = 8192), as S sub-batches on S streams (default 2: two dependency chains fill each other's
//      memory-bound passes);
//   2. the hint sources (five tunnels' linear functions and key-switch hints, four quadratic hints -- generated once per circuit,
//      Crypto/Alchemy/Interpreter/KeysHints.hs:101-129) are generated on rank 0 ONLY and broadcast with alch_hint_broadcast; the
//      other ranks' sources start zeroed, so a broadcast that did not deliver cannot go unnoticed;
//   3. every rank runs K timed passes of the op sequence on its shard between two barriers -- no collective inside;
//   4. every shard's result is checked against the C restatement's per-ciphertext checksums (tests/golden/batch_checksums.json,
//      "homomrlwr"): the shards carry the same seeded inputs, so the fixture applies to each of them;
//   5. the first G result ciphertexts of every rank are all-gathered (alch_buf_all_gather) and every rank checks every slice.
// Prints one JSON line: aggregate ringRound evaluations per second (N * B / slowest rank's time), per-rank times, check results.
// Exit status 0 only when every check passed.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "../alchemy_amd/host/ringround.hpp"
#include "../include/alchemy_rccl.h"

using alchemy::ringround::Lanes;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the "per_ciphertext" list of the "homomrlwr" section of tests/golden/batch_checksums.json
static std::vector<uint64_t> load_fixture(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    size_t a = s.find("\"homomrlwr\"");
    if (a == std::string::npos) throw std::runtime_error("no homomrlwr section in " + path);
    a = s.find("\"per_ciphertext\"", a);
    a = s.find('[', a);
    const size_t b = s.find(']', a);
    std::vector<uint64_t> out;
    for (size_t i = a; i < b;) {
        const size_t q0 = s.find('"', i);
        if (q0 == std::string::npos || q0 > b) break;
        const size_t q1 = s.find('"', q0 + 1);
        out.push_back(strtoull(s.substr(q0 + 1, q1 - q0 - 1).c_str(), nullptr, 16));
        i = q1 + 1;
    }
    return out;
}

struct Barrier {
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const int gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_; });
    }
    std::mutex mu_;
    std::condition_variable cv_;
    int n_, count_ = 0, gen_ = 0;
};

int main(int argc, char** argv) {
    int N = 1, passes = 2, S = 2, world = 0, my_rank = -1, device = -1;
    std::string id_file;
    size_t B = 1024, G = 16;
    std::string fixture = "tests/golden/batch_checksums.json";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) N = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--batch") && i + 1 < argc) B = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "--passes") && i + 1 < argc) passes = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--lanes") && i + 1 < argc) S = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gather") && i + 1 < argc) G = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "--fixture") && i + 1 < argc) fixture = argv[++i];
        else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc) my_rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) id_file = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    try {
        if (N < 1 || passes < 1 || B < 1 || S < 1) throw std::runtime_error("--gpus, --passes, --lanes and --batch must be positive");
        S = (int)std::min((size_t)S, B);
        G = std::min(G, B / (size_t)S);                                   // the gathered ciphertexts come from the first sub-batch
        const std::vector<uint64_t> fix = load_fixture(fixture);
        alch_comm* comm = nullptr;
        const bool per_process = world > 0;
        // W ranks in the communicator, NL of them in this process: local rank i is rank rank_of(i) on device dev_of(i)
        const int W = per_process ? world : N, NL = per_process ? 1 : N;
        if (per_process) {
            if (my_rank < 0 || my_rank >= world || id_file.empty()) throw std::runtime_error("--world N needs --rank R (0 <= R < N) and --id-file PATH");
            int visible = 0;
            if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) throw std::runtime_error("no HIP device");
            if (device < 0) device = my_rank % visible;
            if (device >= visible) throw std::runtime_error("--device " + std::to_string(device) + " but only " + std::to_string(visible) + " devices visible");
            if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
            unsigned char id[ALCH_COMM_ID_BYTES];
            if (my_rank == 0) {                                           // the id travels through a file: written whole, then renamed
                if (alch_comm_unique_id(id) != ALCH_OK) throw std::runtime_error(std::string("alch_comm_unique_id: ") + alch_rccl_last_error());
                const std::string tmp = id_file + ".tmp";
                { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char*>(id), sizeof id); if (!f) throw std::runtime_error("cannot write " + tmp); }
                if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
            } else {
                const double t0 = now();
                for (;;) {
                    std::ifstream f(id_file, std::ios::binary);
                    if (f && f.read(reinterpret_cast<char*>(id), sizeof id) && f.gcount() == (std::streamsize)sizeof id) break;
                    if (now() - t0 > 120.0) throw std::runtime_error("no id in " + id_file + " after 120 s: is rank 0 running?");
                    std::this_thread::sleep_for(std::chrono::milliseconds(50));
                }
            }
            if (alch_comm_init_rank(world, my_rank, id, &comm) != ALCH_OK) throw std::runtime_error(std::string("alch_comm_init_rank: ") + alch_rccl_last_error());
        } else if (alch_comm_init_all(N, &comm) != ALCH_OK) throw std::runtime_error(std::string("alch_comm_init_all: ") + alch_rccl_last_error());
        auto rank_of = [&](int i) { return per_process ? my_rank : i; };
        auto dev_of = [&](int i) { return per_process ? device : i; };

        std::vector<std::unique_ptr<Lanes>> rr((size_t)NL);
        std::vector<std::string> errors((size_t)NL);
        std::vector<double> secs((size_t)NL, 0.0);
        std::vector<int> shard_ok((size_t)NL, 0), gather_ok((size_t)NL, 0);
        std::vector<alch_buf*> results((size_t)NL, nullptr), gathered((size_t)NL, nullptr);
        Barrier bar(NL);
        std::atomic<bool> failed(false);
        double t_bcast = 0;

        auto guarded = [&](int r, const std::function<void()>& fn) {
            if (failed) return;
            try { fn(); } catch (const std::exception& e) { errors[(size_t)r] = e.what(); failed = true; }
        };
        auto worker = [&](int r) {
            guarded(r, [&] {
                if (hipSetDevice(dev_of(r)) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
                rr[(size_t)r].reset(new Lanes(B, S));
                if (rank_of(r) == 0) rr[(size_t)r]->fillSources();        // generated once, on one rank
                else {
                    for (auto& ln : rr[(size_t)r]->lane) for (auto& s : ln->sources) {   // zeros until the broadcast arrives
                        void* p = nullptr;
                        size_t bytes = 0;
                        alchemy::ringround::check(alch_buf_device_ptr(s.first, &p, &bytes), "alch_buf_device_ptr");
                        if (hipMemset(p, 0, bytes) != hipSuccess) throw std::runtime_error("hipMemset failed");
                    }
                    if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("hipDeviceSynchronize failed");
                }
                rr[(size_t)r]->sync();
            });
            bar.wait();
            if (r == 0) guarded(0, [&] {                                  // the one collective of the path, before anything is timed
                const double t0 = now();
                const size_t ns = rr[0]->lane[0]->sources.size();
                for (size_t l = 0; l < rr[0]->lane.size(); ++l) for (size_t i = 0; i < ns; ++i) {
                    std::vector<alch_buf*> bufs;
                    size_t elems = 0;
                    for (int k = 0; k < NL; ++k) bufs.push_back(rr[(size_t)k]->lane[l]->sources[i].first);
                    alchemy::ringround::check(alch_buf_elems(bufs[0], &elems), "alch_buf_elems");
                    if (alch_hint_broadcast(comm, 0, bufs.data(), 0, elems) != ALCH_OK) throw std::runtime_error(std::string("alch_hint_broadcast: ") + alch_rccl_last_error());
                }
                for (int k = 0; k < NL; ++k) rr[(size_t)k]->sync();
                t_bcast = now() - t0;
            });
            bar.wait();
            guarded(r, [&] {
                (void)hipSetDevice(dev_of(r));
                rr[(size_t)r]->buildHints();
                rr[(size_t)r]->run();                                     // allocations, first touch
                rr[(size_t)r]->sync();
            });
            bar.wait();
            guarded(r, [&] {
                const double t0 = now();
                std::vector<alch_buf*> outs;
                for (int p = 0; p < passes; ++p) outs = rr[(size_t)r]->run();
                rr[(size_t)r]->sync();
                secs[(size_t)r] = (now() - t0) / passes;
                alch_buf* res = outs[0];
                results[(size_t)r] = res;
                // the shard against the oracle: per-ciphertext checksums are position dependent, so any prefix adds up
                const size_t cnt = std::min(B, fix.size());
                uint64_t want = 0, got = 0;
                for (size_t i = 0; i < cnt; ++i) want += fix[i];
                got = rr[(size_t)r]->checksum(outs, cnt);
                shard_ok[(size_t)r] = got == want;
                alch_ring* ring = nullptr;
                alchemy::ringround::check(alch_buf_ring(res, &ring), "alch_buf_ring");
                alchemy::ringround::check(alch_buf_alloc(ring, (size_t)W * 2 * G, &gathered[(size_t)r]), "alch_buf_alloc");
            });
            bar.wait();
            if (r == 0) guarded(0, [&] {                                  // the batch gather, after timing
                if (alch_buf_all_gather(comm, results.data(), 0, 2 * G, gathered.data()) != ALCH_OK)
                    throw std::runtime_error(std::string("alch_buf_all_gather: ") + alch_rccl_last_error());
            });
            bar.wait();
            guarded(r, [&] {
                // every slice equals this rank's own first G ciphertexts (the shards carry the same data) -- bit for bit
                uint64_t own = 0;
                alchemy::ringround::check(alch_buf_checksum(results[(size_t)r], 0, 2 * G, &own), "alch_buf_checksum");
                bool ok = true;
                for (int k = 0; k < W; ++k) {
                    uint64_t got = 0;
                    alchemy::ringround::check(alch_buf_checksum(gathered[(size_t)r], (size_t)k * 2 * G, 2 * G, &got), "alch_buf_checksum");
                    ok = ok && got == own;
                }
                gather_ok[(size_t)r] = ok;
            });
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < NL; ++r) threads.emplace_back(worker, r);
        for (auto& t : threads) t.join();
        for (alch_buf* g : gathered) if (g) alch_buf_free(g);
        rr.clear();
        alch_comm_destroy(comm);
        bool ok = !failed;
        double slow = 0;
        for (int r = 0; r < NL; ++r) { ok = ok && shard_ok[(size_t)r] && gather_ok[(size_t)r]; slow = std::max(slow, secs[(size_t)r]); }
        // one process per GPU: pipelines_per_s is THIS process's rate (the launcher takes ranks * B / the slowest rank's time)
        printf("{\"workload\": \"HomomRLWR ringRound pipeline, %zu ciphertexts per GPU as %d sub-batches, native host (C++ %s + RCCL, no torch)\", \"n_gpus\": %d, "
               "\"ranks_in_this_process\": %d, \"first_rank\": %d, "
               "\"pipelines_per_s\": %.1f, \"ms_per_pass_slowest_rank\": %.3f, \"hint_broadcast_ms\": %.3f, \"shard_checksums_ok\": [", B, S,
               per_process ? "processes" : "threads", W, NL, rank_of(0), slow > 0 ? (double)NL * (double)B / slow : 0.0, slow * 1e3, t_bcast * 1e3);
        for (int r = 0; r < NL; ++r) printf("%s%s", r ? ", " : "", shard_ok[(size_t)r] ? "true" : "false");
        printf("], \"all_gather_slices_ok\": [");
        for (int r = 0; r < NL; ++r) printf("%s%s", r ? ", " : "", gather_ok[(size_t)r] ? "true" : "false");
        printf("], \"gathered_ciphertexts_per_rank\": %zu, \"ciphertexts_checked_per_shard\": %zu}\n", G, std::min(B, fix.size()));
        for (int r = 0; r < NL; ++r) if (!errors[(size_t)r].empty()) fprintf(stderr, "rank %d: %s\n", rank_of(r), errors[(size_t)r].c_str());
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}

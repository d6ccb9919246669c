// trc_ranks -- N ranks of the path-tracing hot path started and composed WITHOUT Python or PyTorch (north_star: "Host code
// stays in the repo's own Swift/C++ calling HIP through a thin C-ABI/FFI layer (no PyTorch)").
//
//   trc_ranks --ranks N [--host-collectives] [--samples] [--size W H] [--spp S] [--sppm FRAMES] [--out frame.png]
//
// Without TRC_RANK in the environment the program is the LAUNCHER: it forks N children before anything has touched the
// GPU (no exec: a child simply goes on as rank r of this same image; HIP is first used after the fork, in the child) and
// waits for them.  A launcher of your own (mpirun, srun, a shell loop) sets TRC_RANK / TRC_WORLD / TRC_PORT instead.  A rank then
//   1. meets the others on 127.0.0.1:TRC_PORT (rank 0 listens, the others connect: one TCP connection per rank),
//   2. joins the group: by default rank 0 makes the RCCL unique id (trc_group_unique_id) and sends its 128 bytes down the
//      sockets, every rank calls trc_group_init -- one GPU per rank, RCCL over xGMI; with --host-collectives the
//      sockets themselves carry the collectives through trc_group_set_collectives (host-staged, a star through rank 0),
//      which is how N ranks run on ONE GPU (RCCL refuses a second rank on a device),
//      The choice is made from the devices the ranks HOLD: every rank sends the PCI bus id of its context's device up the star
//      (trc_device_pci_bus_id); two ranks on one device -> host collectives, whatever the command line asked for,
//   3. renders its tiles (tx + ty) % N == rank of the BASELINE config-2 frame, composes with trc_group_reduce_accum, and --
//      with --sppm F -- runs F frames of the SPPM pass with its all-reduce / all-gather,
//   4. rank 0 renders the same frame alone and checks that the composed frame equals it bit for bit, writes the PNG.
// --samples: the split that scales (tracer_abi.h "sample sharding") instead of step 3's tiles: every rank the WHOLE frame with
// spp / N samples from trc_seed(trc_shard_seed(seed, rank)), composed by trc_group_compose_samples (all-to-all of pixel slices,
// rank-ordered fold, gather); rank 0 then renders the N shards itself, folds them in rank order on the host and checks the bits.
// The reference has no counterpart (its multi-device path is commented out, AAPLRenderer.mm:139-146).
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tracer_abi.h"

namespace {

bool send_all(int fd, const void* p, size_t n) {
    const char* b = static_cast<const char*>(p);
    while (n) { const ssize_t k = ::send(fd, b, n, MSG_NOSIGNAL); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}
bool recv_all(int fd, void* p, size_t n) {
    char* b = static_cast<char*>(p);
    while (n) { const ssize_t k = ::recv(fd, b, n, 0); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}

// the star: rank 0 holds one socket per other rank (peers[r]), every other rank one socket to rank 0 (peers[0])
struct Star {
    int rank = 0, world = 1;
    std::vector<int> peers;
    std::vector<char> scratch;
};

template <typename T, typename Op>
void combine(T* acc, const T* in, size_t n, Op op) { for (size_t i = 0; i < n; ++i) acc[i] = op(acc[i], in[i]); }

bool reduce_into_root(Star& s, void* buf, size_t count, int dtype, int op) {
    const size_t bytes = count * (dtype == TRC_DT_U8 ? 1 : 4);
    if (s.rank != 0) return send_all(s.peers[0], buf, bytes);
    s.scratch.resize(bytes);
    for (int r = 1; r < s.world; ++r) {                      // rank order: the float sum is a fixed left-to-right chain
        if (!recv_all(s.peers[r], s.scratch.data(), bytes)) return false;
        if (dtype == TRC_DT_F32 && op == TRC_OP_SUM) combine((float*)buf, (const float*)s.scratch.data(), count, [](float a, float b) { return a + b; });
        else if (dtype == TRC_DT_U32 && op == TRC_OP_MIN) combine((uint32_t*)buf, (const uint32_t*)s.scratch.data(), count, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
        else if (dtype == TRC_DT_U32 && op == TRC_OP_MAX) combine((uint32_t*)buf, (const uint32_t*)s.scratch.data(), count, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
        else if (dtype == TRC_DT_U32 && op == TRC_OP_SUM) combine((uint32_t*)buf, (const uint32_t*)s.scratch.data(), count, [](uint32_t a, uint32_t b) { return a + b; });
        else return false;
    }
    return true;
}
bool broadcast_from_root(Star& s, void* buf, size_t bytes) {
    if (s.rank != 0) return recv_all(s.peers[0], buf, bytes);
    for (int r = 1; r < s.world; ++r) if (!send_all(s.peers[r], buf, bytes)) return false;
    return true;
}
// trc_collectives over the star (host_staged = 1: `buf` is host memory, `stream` NULL)
int star_reduce(void* user, void* buf, size_t count, int dtype, int op, int root, void*) {
    Star& s = *static_cast<Star*>(user);
    if (root != 0) return 2;                                 // this example composes on rank 0
    return reduce_into_root(s, buf, count, dtype, op) ? 0 : 1;
}
int star_allreduce(void* user, void* buf, size_t count, int dtype, int op, void*) {
    Star& s = *static_cast<Star*>(user);
    return reduce_into_root(s, buf, count, dtype, op) && broadcast_from_root(s, buf, count * (dtype == TRC_DT_U8 ? 1 : 4)) ? 0 : 1;
}
int star_allgather(void* user, void* buf, size_t bytes_per_rank, void*) {
    Star& s = *static_cast<Star*>(user);
    char* b = static_cast<char*>(buf);
    if (s.rank != 0) { if (!send_all(s.peers[0], b + (size_t)s.rank * bytes_per_rank, bytes_per_rank)) return 1; }
    else for (int r = 1; r < s.world; ++r) if (!recv_all(s.peers[r], b + (size_t)r * bytes_per_rank, bytes_per_rank)) return 1;
    return broadcast_from_root(s, buf, bytes_per_rank * (size_t)s.world) ? 0 : 1;
}

// sample-sharded compose: slice p of every rank's buffer goes to rank p (through the hub), then the composed slices to the root
int star_alltoall(void* user, void* buf, size_t bytes_per_rank, void*) {
    Star& s = *static_cast<Star*>(user);
    char* b = static_cast<char*>(buf);
    const size_t all = bytes_per_rank * (size_t)s.world;
    if (s.rank != 0) return send_all(s.peers[0], b, all) && recv_all(s.peers[0], b, all) ? 0 : 1;
    std::vector<std::vector<char>> rows((size_t)s.world);             // rows[r] = rank r's buffer
    rows[0].assign(b, b + all);
    for (int r = 1; r < s.world; ++r) { rows[(size_t)r].resize(all); if (!recv_all(s.peers[r], rows[(size_t)r].data(), all)) return 1; }
    std::vector<char> out(all);
    for (int r = s.world - 1; r >= 0; --r) {                           // what rank r receives: slice r of every row
        for (int p = 0; p < s.world; ++p) std::memcpy(out.data() + (size_t)p * bytes_per_rank, rows[(size_t)p].data() + (size_t)r * bytes_per_rank, bytes_per_rank);
        if (r == 0) std::memcpy(b, out.data(), all);
        else if (!send_all(s.peers[r], out.data(), all)) return 1;
    }
    return 0;
}
int star_gather(void* user, void* buf, size_t bytes_per_rank, int root, void*) {
    Star& s = *static_cast<Star*>(user);
    if (root != 0) return 2;                                 // this example composes on rank 0
    char* b = static_cast<char*>(buf);
    if (s.rank != 0) return send_all(s.peers[0], b + (size_t)s.rank * bytes_per_rank, bytes_per_rank) ? 0 : 1;
    for (int r = 1; r < s.world; ++r) if (!recv_all(s.peers[r], b + (size_t)r * bytes_per_rank, bytes_per_rank)) return 1;
    return 0;
}

bool meet(Star& s, int port) {
    s.peers.assign((size_t)s.world, -1);
    sockaddr_in addr{};
    addr.sin_family = AF_INET; addr.sin_port = htons((uint16_t)port); addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    const int one = 1;
    if (s.rank == 0) {
        const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        if (::bind(ls, (sockaddr*)&addr, sizeof addr) != 0 || ::listen(ls, s.world) != 0) { std::perror("bind/listen"); return false; }
        for (int k = 1; k < s.world; ++k) {
            const int fd = ::accept(ls, nullptr, nullptr);
            int32_t who = -1;
            if (fd < 0 || !recv_all(fd, &who, 4) || who <= 0 || who >= s.world || s.peers[(size_t)who] != -1) return false;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            s.peers[(size_t)who] = fd;
        }
        ::close(ls);
    } else {
        const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        for (int tries = 0; ::connect(fd, (sockaddr*)&addr, sizeof addr) != 0; ++tries) {
            if (tries > 600) { std::perror("connect"); return false; }       // rank 0 may still be starting
            ::usleep(50 * 1000);
        }
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        const int32_t who = s.rank;
        if (!send_all(fd, &who, 4)) return false;
        s.peers[0] = fd;
    }
    return true;
}

int free_port() {
    const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
    socklen_t len = sizeof a;
    int port = 29599;
    if (::bind(fd, (sockaddr*)&a, sizeof a) == 0 && ::getsockname(fd, (sockaddr*)&a, &len) == 0) port = ntohs(a.sin_port);
    ::close(fd);
    return port;
}

#define CHECK(call)                                                                                              \
    do {                                                                                                         \
        const trc_status st_ = (call);                                                                           \
        if (st_ != TRC_OK) { std::fprintf(stderr, "rank %d: %s failed: %s (%s)\n", star.rank, #call, trc_status_string(st_), ctx ? trc_last_error(ctx) : ""); return 1; } \
    } while (0)

}  // namespace

int main(int argc, char** argv) {
    int ranks = 2, sppm_frames = 0;
    uint32_t W = 1920, H = 1080, spp = 64;
    bool host_collectives = false, samples = false;
    std::string out = "ranks.png";
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--ranks" && i + 1 < argc) ranks = std::atoi(argv[++i]);
        else if (a == "--host-collectives") host_collectives = true;
        else if (a == "--samples") samples = true;
        else if (a == "--size" && i + 2 < argc) { W = (uint32_t)std::atoi(argv[++i]); H = (uint32_t)std::atoi(argv[++i]); }
        else if (a == "--spp" && i + 1 < argc) spp = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--sppm" && i + 1 < argc) sppm_frames = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (ranks < 1 || ranks > 64) { std::fprintf(stderr, "--ranks 1..64\n"); return 2; }

    Star star;
    int port = 29599;
    if (const char* env_rank = std::getenv("TRC_RANK")) {     // started by somebody else's launcher
        star.rank = std::atoi(env_rank);
        star.world = std::atoi(std::getenv("TRC_WORLD") ? std::getenv("TRC_WORLD") : "1");
        port = std::atoi(std::getenv("TRC_PORT") ? std::getenv("TRC_PORT") : "29599");
    } else {                                                  // ---- launcher: nothing here touches the GPU
        port = free_port();
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < ranks && !child; ++r) {
            const pid_t pid = ::fork();
            if (pid < 0) { std::perror("fork"); return 1; }
            if (pid == 0) { child = true; star.rank = r; star.world = ranks; ::setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); }
            else kids.push_back(pid);
        }
        if (!child) {
            int worst = 0;
            for (pid_t pid : kids) {                          // our own children, by PID
                int st = 0;
                ::waitpid(pid, &st, 0);
                const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
                if (rc > worst) worst = rc;
            }
            return worst;
        }
    }

    // ---- one rank
    trc_ctx* ctx = nullptr;
    if (!meet(star, port)) { std::fprintf(stderr, "rank %d: rendezvous failed\n", star.rank); return 1; }

    trc_host_scene* hs = nullptr;
    if (trc_host_scene_create(TRC_SCENE_CORNELL_SPHERES, nullptr, 0, nullptr, 0, &hs) != TRC_OK) return 1;
    trc_scene scene;
    trc_host_scene_view(hs, &scene);
    trc_Camera cam;
    trc_host_prepare_camera(&cam, (float)W, (float)H);

    // one GPU per rank when there are enough of them, else the ranks share -- which the bus ids below find out
    int device = 0;
    for (int d = star.rank; d >= 0; --d) { if (trc_create(d, &ctx) == TRC_OK) { device = d; break; } ctx = nullptr; }
    if (!ctx) { std::fprintf(stderr, "rank %d: no GPU\n", star.rank); return 1; }
    CHECK(trc_upload_scene(ctx, &scene));
    CHECK(trc_set_camera(ctx, &cam));
    CHECK(trc_resize(ctx, W, H));
    {   // RCCL only when every rank holds a device of its own: decided from the PCI bus ids, by rank 0, for everybody
        char bus[64] = {0};
        CHECK(trc_device_pci_bus_id(ctx, bus, sizeof bus));
        std::vector<char> all((size_t)star.world * 64, 0);
        std::memcpy(all.data() + (size_t)star.rank * 64, bus, 64);
        if (star_allgather(&star, all.data(), 64, nullptr) != 0) return 1;
        bool shared = false;
        for (int i = 0; i < star.world; ++i)
            for (int j = i + 1; j < star.world; ++j) shared |= std::strncmp(all.data() + (size_t)i * 64, all.data() + (size_t)j * 64, 64) == 0;
        if (shared && !host_collectives) {
            if (star.rank == 0) std::fprintf(stderr, "ranks share a device (%s ...): RCCL refuses a second rank on one, composing over the sockets\n", all.data());
            host_collectives = true;
        }
    }

    trc_collectives table{};
    if (host_collectives) {
        table.user = &star; table.host_staged = 1;
        table.reduce = star_reduce; table.allreduce = star_allreduce; table.allgather = star_allgather;
        table.alltoall = star_alltoall; table.gather = star_gather;
        CHECK(trc_group_set_collectives(ctx, &table, star.world, star.rank));
    } else {
        uint8_t id[TRC_UNIQUE_ID_BYTES] = {0};
        if (star.rank == 0 && trc_group_unique_id(id) != TRC_OK) { std::fprintf(stderr, "trc_group_unique_id failed (no RCCL?)\n"); return 1; }
        if (!broadcast_from_root(star, id, sizeof id)) return 1;                // the id travels out of band: 128 bytes down the sockets
        CHECK(trc_group_init(ctx, id, star.world, star.rank));
    }

    trc_params prm;
    std::memset(&prm, 0, sizeof prm);
    prm.spp = spp; prm.max_depth = 8; prm.integrator = TRC_INTEGRATOR_PATH;
    prm.tile_rank = (uint32_t)star.rank; prm.tile_nranks = (uint32_t)star.world;
    if (samples) {
        if (spp % (uint32_t)star.world) { std::fprintf(stderr, "--samples: %u samples do not split over %d ranks\n", spp, star.world); return 2; }
        prm.spp = spp / (uint32_t)star.world; prm.tile_rank = 0; prm.tile_nranks = 1;      // the whole frame, this rank's share of the samples
    }
    const size_t n = (size_t)W * H * 4;
    std::vector<float> composed(n), alone(n);
    double ms[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {                    // second pass: adaptive order and block sizes, warm collectives
        CHECK(trc_clear_accum(ctx));
        CHECK(trc_seed(ctx, samples ? trc_shard_seed(0x5EED0000ull, (uint32_t)star.rank) : 0x5EED0000ull));
        CHECK(trc_synchronize(ctx));
        int32_t go = 1;
        if (!broadcast_from_root(star, &go, 4)) return 1;     // a cheap barrier: everybody starts the timed pass together
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(trc_render(ctx, &prm));
        if (samples) CHECK(trc_group_compose_samples(ctx, 0, 0));
        else CHECK(trc_group_reduce_accum(ctx, 0));
        CHECK(trc_synchronize(ctx));
        ms[pass] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    trc_stats st;
    CHECK(trc_get_stats(ctx, &st));
    if (star.rank == 0) { if (samples) CHECK(trc_download_composed(ctx, composed.data())); else CHECK(trc_download_accum(ctx, composed.data())); }

    int bad = 0;
    if (sppm_frames > 0) {                                    // the grouped SPPM pass: all-reduce of the bounds, all-gather of the photons
        CHECK(trc_clear_accum(ctx));
        CHECK(trc_seed(ctx, 8));
        CHECK(trc_sppm_init(ctx, 9));
        CHECK(trc_sppm_frames(ctx, (uint32_t)sppm_frames));
        CHECK(trc_group_reduce_accum(ctx, 0));
        std::vector<float> sppm_n(n), sppm_1(n);
        if (star.rank == 0) CHECK(trc_download_accum(ctx, sppm_n.data()));
        CHECK(trc_group_finalize(ctx));
        if (star.rank == 0) {
            CHECK(trc_clear_accum(ctx)); CHECK(trc_seed(ctx, 8)); CHECK(trc_sppm_init(ctx, 9)); CHECK(trc_sppm_frames(ctx, (uint32_t)sppm_frames));
            CHECK(trc_download_accum(ctx, sppm_1.data()));
            const bool same = std::memcmp(sppm_n.data(), sppm_1.data(), n * 4) == 0;
            std::printf("SPPM, %d frames: the %d-rank frame %s the 1-rank frame\n", sppm_frames, star.world, same ? "==" : "DIFFERS FROM");
            bad |= !same;
        }
    } else {
        CHECK(trc_group_finalize(ctx));
    }
    if (star.rank == 0) {                                     // the same frame on one rank: must be the same bits
        prm.tile_rank = 0; prm.tile_nranks = 1;
        if (samples) {                                        // the definition, by hand: the N shards one after the other, folded in rank order
            std::vector<float> shard(n);
            for (int g = 0; g < star.world; ++g) {
                CHECK(trc_clear_accum(ctx)); CHECK(trc_seed(ctx, trc_shard_seed(0x5EED0000ull, (uint32_t)g))); CHECK(trc_render(ctx, &prm));
                CHECK(trc_download_accum(ctx, shard.data()));
                if (g == 0) alone = shard;
                else for (size_t i = 0; i < n; ++i) alone[i] += shard[i];
            }
            const float groups = (float)star.world;
            for (size_t i = 0; i < n; ++i) alone[i] /= groups;
        } else {
            CHECK(trc_clear_accum(ctx)); CHECK(trc_seed(ctx, 0x5EED0000ull)); CHECK(trc_render(ctx, &prm));
            CHECK(trc_download_accum(ctx, alone.data()));
        }
        const bool same = std::memcmp(composed.data(), alone.data(), n * 4) == 0;
        std::printf("%d ranks (%s), %ux%ux%uspp tracePath%s: rank 0 rendered %llu rays per pass, step %.2f ms (first %.2f); composed frame %s %s\n",
                    star.world, host_collectives ? "collectives over TCP sockets, host-staged" : "RCCL", W, H, spp,
                    samples ? ", samples split over the ranks' seeds" : "",
                    (unsigned long long)(st.rays / 2), ms[1], ms[0], same ? "==" : "DIFFERS FROM",
                    samples ? "the rank-ordered mean of the shards rendered on one rank" : "the 1-rank frame");
        bad |= !same;
        std::vector<uint8_t> rgba8((size_t)W * H * 4);
        CHECK(trc_upload_accum(ctx, composed.data()));
        CHECK(trc_tonemap(ctx, rgba8.data(), nullptr));
        if (trc_host_write_png(out.c_str(), rgba8.data(), W, H) != TRC_OK) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); bad = 1; }
    }
    int32_t verdict = bad;
    if (!broadcast_from_root(star, &verdict, 4)) return 1;    // every rank leaves with rank 0's verdict
    (void)device;
    trc_destroy(ctx);
    trc_host_scene_destroy(hs);
    for (int fd : star.peers) if (fd >= 0) ::close(fd);
    return verdict ? 1 : 0;
}

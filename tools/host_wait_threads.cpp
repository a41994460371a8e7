// Eight handles driven by eight HOST THREADS through the C ABI — the shape of an 8-rank node's host load — under the
// three host-wait policies of gsr_host_wait_policy (gsr.h): pure spin, the default (30 us spin, then sched_yield polling)
// and the opt-in adaptive sleep.  Native on purpose: the same experiment from eight Python threads measures the GIL
// (0.8 .. 2.2 ms per step for one and the same policy), not the wait.
//
//   hipcc -O2 -std=c++17 tools/host_wait_threads.cpp -Iinclude -Lgaussiansplatting.jl_amd -lgsr_hip
//         -Wl,-rpath,'$ORIGIN' -o gaussiansplatting.jl_amd/host_wait_threads        (what __graft_entry__.build() runs)
//   host_wait_threads [threads = 8] [steps = 300] [rounds = 3]
// Prints one line per policy (best round: wall ms per step, process CPU ms per step, checksum of the images) and
//   RESULT spin <ms> default <ms> sleep <ms> images_equal <0|1>
#include <hip/hip_runtime.h>
#include <sys/resource.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gsr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define GK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "gsr error %d (%s) at %s:%d\n", r_, gsr_last_error_string(), __FILE__, __LINE__); exit(3); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {  // splitmix64 -> [0, 1)
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static double nrand() { const double u = urand() + 1e-300, v = urand(); return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }

template <class T> static T* upload(const std::vector<T>& v) {
    T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}
static double cpu_seconds() {
    rusage ru; getrusage(RUSAGE_SELF, &ru);
    return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
}

int main(int argc, char** argv) {
    // a stale binary (built against another include/gsr.h) must fail here, not corrupt its stack through a grown struct
    if (gsr_check_abi(GSR_ABI_VERSION, sizeof(gsr_config), sizeof(gsr_inputs), sizeof(gsr_camera), sizeof(gsr_aux), sizeof(gsr_stats),
                      sizeof(gsr_grads), sizeof(gsr_tail_state)) != 0) {
        fprintf(stderr, "%s\n", gsr_last_error_string());
        return 3;
    }
    const int T = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 300, rounds = argc > 3 ? atoi(argv[3]) : 3;
    const int W = 640, H = 480, N = 60000, K = 4, deg = 1;
    if (gsr_abi_version() != GSR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 4; }
    // the synthetic scene of SURVEY.md §8d (camera at the origin looking down +z, 60 degree horizontal field of view)
    const double fx = 0.5 * W / std::tan(M_PI / 6.0);
    std::vector<float> means(3 * N), shs(3 * K * N), opac(N), scales(3 * N), rots(4 * N);
    for (int i = 0; i < N; i++) {
        const double z = 2.0 + 10.0 * urand(), u = 2.0 * urand() - 1.0, v = 2.0 * urand() - 1.0;
        means[3 * i] = (float)(1.1 * z * u * W / (2.0 * fx)); means[3 * i + 1] = (float)(1.1 * z * v * H / (2.0 * fx)); means[3 * i + 2] = (float)z;
        const double n0 = nrand();
        for (int k = 0; k < 3; k++) scales[3 * i + k] = (float)std::exp(std::log(3.0 * z / fx) + 0.35 * n0 + 0.3 * nrand());
        for (int k = 0; k < 4; k++) rots[4 * i + k] = (float)nrand();
        opac[i] = (float)(1.0 / (1.0 + std::exp(-(nrand() - 1.0))));
        for (int k = 0; k < 3 * K; k++) shs[(size_t)3 * K * i + k] = (float)((k < 3 ? 0.5 : 0.1) * nrand());
    }
    std::vector<float> vp((size_t)3 * W * H);
    for (auto& x : vp) x = (float)(nrand() / (3.0 * W * H));
    gsr_inputs in{};
    in.n = N; in.n_coeffs = K; in.sh_degree = deg;
    in.means = upload(means); in.shs = upload(shs); in.opacities = upload(opac); in.scales = upload(scales); in.rotations = upload(rots);
    const float* vpix = upload(vp);
    gsr_camera cam{};
    cam.R[0] = cam.R[4] = cam.R[8] = 1.0f;
    cam.focal[0] = cam.focal[1] = (float)fx; cam.principal[0] = cam.principal[1] = 0.5f;

    struct Slot { gsr_handle* h; hipStream_t s; float* image; gsr_grads g; };
    std::vector<Slot> slot(T);
    gsr_config cfg{};
    cfg.width = W; cfg.height = H; cfg.mode = GSR_MODE_RGB; cfg.near_plane = 0.2f; cfg.far_plane = 1000.0f; cfg.radius_clip = 3; cfg.blur_eps = 0.3f;
    cfg.ssim_precision = GSR_DEFAULT; cfg.preprocess_form = GSR_DEFAULT;
    for (auto& sl : slot) {
        GK(gsr_create(&cfg, &sl.h));
        CK(hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking));
        CK(hipMalloc(&sl.image, sizeof(float) * 3 * W * H));
        sl.g = gsr_grads{};
        CK(hipMalloc(&sl.g.vmeans, sizeof(float) * 3 * N)); CK(hipMalloc(&sl.g.vshs, sizeof(float) * 3 * K * N));
        CK(hipMalloc(&sl.g.vopacities, sizeof(float) * N)); CK(hipMalloc(&sl.g.vscales, sizeof(float) * 3 * N));
        CK(hipMalloc(&sl.g.vrotations, sizeof(float) * 4 * N));
    }
    auto step = [&](Slot& sl) {
        GK(gsr_forward(sl.h, &in, &cam, sl.image, nullptr, sl.s, nullptr));
        GK(gsr_backward(sl.h, &in, &cam, vpix, &sl.g, sl.s));
    };
    auto run = [&](int spin, int yield, int sleep, double& wall_ms, double& cpu_ms, double& checksum) {
        GK(gsr_host_wait_policy(spin, yield, sleep));
        std::atomic<int> ready{0}; std::atomic<bool> go{false};
        std::vector<std::thread> th;
        for (int i = 0; i < T; i++)
            th.emplace_back([&, i] {
                CK(hipSetDevice(0));
                for (int k = 0; k < 5; k++) step(slot[i]);
                CK(hipStreamSynchronize(slot[i].s));
                ready++;
                while (!go.load()) std::this_thread::yield();
                for (int k = 0; k < steps; k++) step(slot[i]);
                CK(hipStreamSynchronize(slot[i].s));
            });
        while (ready.load() < T) std::this_thread::yield();
        const double c0 = cpu_seconds();
        const auto t0 = std::chrono::steady_clock::now();
        go = true;
        for (auto& t : th) t.join();
        wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
        cpu_ms = (cpu_seconds() - c0) * 1e3 / steps;
        std::vector<float> img((size_t)3 * W * H);
        checksum = 0.0;
        for (auto& sl : slot) {
            CK(hipMemcpy(img.data(), sl.image, img.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < img.size(); k += 7) checksum += img[k] * (double)(1 + k % 13);
        }
    };
    const int pol[3][3] = {{1000000, 0, 0}, {30, 0, 0}, {100, 0, 50}};
    const char* name[3] = {"pure spin", "default (30 us spin, then sched_yield)", "adaptive sleep (100, 0, 50)"};
    double best[3] = {1e30, 1e30, 1e30}, cpu[3] = {0, 0, 0}, sum[3] = {0, 0, 0};
    for (int r = 0; r < rounds; r++)
        for (int p = 0; p < 3; p++) {
            double w, c, s;
            run(pol[p][0], pol[p][1], pol[p][2], w, c, s);
            if (w < best[p]) { best[p] = w; cpu[p] = c; }
            sum[p] = s;
        }
    GK(gsr_host_wait_policy(30, 0, 0));
    for (int p = 0; p < 3; p++)
        printf("%-42s %.4f ms per step (%d threads x 1 view each), %.3f CPU-ms per step, image checksum %.9g\n", name[p], best[p], T, cpu[p], sum[p]);
    printf("RESULT spin %.4f default %.4f sleep %.4f images_equal %d\n", best[0], best[1], best[2], (int)(sum[0] == sum[1] && sum[0] == sum[2]));
    for (auto& sl : slot) GK(gsr_destroy(sl.h));
    return 0;
}

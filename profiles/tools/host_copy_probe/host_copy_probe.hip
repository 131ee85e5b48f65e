// host_copy_probe -- what the pageable host-pointer path of libmpcx pays for: times, for a 16 MB array,
//   memcpy pageable -> page-locked, page-locked -> pageable (fresh and touched pages, 1 and N threads),
//   hipMemcpyAsync H2D / D2H on page-locked memory, hipMemcpy straight from / to pageable memory (the runtime's own
//   staging), hipHostRegister + DMA + hipHostUnregister of a pageable array.
// Build: hipcc -O2 --offload-arch=gfx950 -pthread host_copy_probe.hip -o host_copy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void par_copy(char *d, const char *s, size_t n, int nt)
{
    std::vector<std::thread> th;
    size_t sl = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (int i = 1; i < nt; ++i) { size_t o = i * sl; if (o < n) th.emplace_back([=] { memcpy(d + o, s + o, n - o < sl ? n - o : sl); }); }
    memcpy(d, s, sl < n ? sl : n);
    for (auto &t : th) t.join();
}

static void *fresh(size_t n) { return mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); }

int main()
{
    const size_t N = 16u << 20;
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    void *pin = nullptr, *pin2 = nullptr, *dev = nullptr;
    CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
    CK(hipHostMalloc(&pin2, N, hipHostMallocNonCoherent));
    CK(hipMalloc(&dev, N));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char *page = (char *)malloc(N); memset(page, 1, N); memset(pin, 2, N); memset(pin2, 3, N);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); memcpy(pin, page, N); double t1 = now();
        printf("memcpy pageable -> pinned(default)      %.3f ms\n", (t1 - t0) * 1e3);
        t0 = now(); memcpy(pin2, page, N); t1 = now();
        printf("memcpy pageable -> pinned(noncoherent)  %.3f ms\n", (t1 - t0) * 1e3);
        t0 = now(); memcpy(page, pin, N); t1 = now();
        printf("memcpy pinned(default) -> touched pages %.3f ms\n", (t1 - t0) * 1e3);
        t0 = now(); memcpy(page, pin2, N); t1 = now();
        printf("memcpy pinned(noncoh)  -> touched pages %.3f ms\n", (t1 - t0) * 1e3);
        for (int nt : {1, 4, 8}) {
            char *f = (char *)fresh(N);
            t0 = now(); par_copy(f, (const char *)pin, N, nt); t1 = now();
            printf("copy pinned -> FRESH pages, %d thread(s)  %.3f ms\n", nt, (t1 - t0) * 1e3);
            munmap(f, N);
            t0 = now(); par_copy(page, (const char *)pin, N, nt); t1 = now();
            printf("copy pinned -> touched pages, %d thread(s) %.3f ms\n", nt, (t1 - t0) * 1e3);
        }
        t0 = now(); CK(hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); t1 = now();
        printf("H2D pinned                               %.3f ms\n", (t1 - t0) * 1e3);
        t0 = now(); CK(hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); t1 = now();
        printf("D2H pinned                               %.3f ms\n", (t1 - t0) * 1e3);
        t0 = now(); CK(hipMemcpy(dev, page, N, hipMemcpyHostToDevice)); t1 = now();
        printf("hipMemcpy H2D from pageable              %.3f ms\n", (t1 - t0) * 1e3);
        t0 = now(); CK(hipMemcpy(page, dev, N, hipMemcpyDeviceToHost)); t1 = now();
        printf("hipMemcpy D2H to pageable (touched)      %.3f ms\n", (t1 - t0) * 1e3);
        { char *f = (char *)fresh(N);
          t0 = now(); CK(hipMemcpy(f, dev, N, hipMemcpyDeviceToHost)); t1 = now();
          printf("hipMemcpy D2H to pageable (FRESH)        %.3f ms\n", (t1 - t0) * 1e3); munmap(f, N); }
        { char *f = (char *)fresh(N);
          t0 = now(); CK(hipHostRegister(f, N, hipHostRegisterDefault)); double t2 = now();
          CK(hipMemcpyAsync(f, dev, N, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double t3 = now();
          CK(hipHostUnregister(f)); t1 = now();
          printf("register FRESH %.3f + D2H %.3f + unregister %.3f ms\n", (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t1 - t3) * 1e3); munmap(f, N); }
        { hipPointerAttribute_t at; t0 = now(); for (int i = 0; i < 100; ++i) { if (hipPointerGetAttributes(&at, page + 4096 * i) != hipSuccess) (void)hipGetLastError(); } t1 = now();
          printf("hipPointerGetAttributes on pageable      %.4f ms each\n", (t1 - t0) * 10); }
        printf("--\n");
    }
    return 0;
}

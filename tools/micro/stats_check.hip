// tools/micro/stats_check.hip -- csrc/lrc_stats.h (mean / std per segment, numpy's arithmetic, one workgroup per 8192-chunk)
// against a literal C++ restatement of numpy's pairwise summation, float and double, segment lengths on every branch.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/stats_check.hip -o build_variants/stats_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
namespace {
#include "../../indoor-point-cloud-datasets-controllable-generation-method-for-mobile-robots-3d-scene-perception_amd/csrc/lrc_stats.h"
}
template <typename T> static T pairwise(const T* a, size_t n) {
    if (n < 8) { T r = 0; for (size_t i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        T r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        size_t i;
        for (i = 8; i < n - n % 8; i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    size_t n2 = n / 2; n2 -= n2 % 8;
    return pairwise(a, n2) + pairwise(a + n2, n - n2);
}
template <typename T> static T np_sum(const T* a, size_t n) {
    T total = 0; bool first = true;
    for (size_t c = 0; c < n; c += 8192) { T s = pairwise(a + c, std::min<size_t>(8192, n - c)); total = first ? s : total + s; first = false; }
    return total;
}
template <typename T> static int run(const char* name) {
    std::vector<uint64_t> counts = {134, 0, 5, 128, 129, 8192, 8193, 70000, 1, 16384, 65536, 40000, 7, 8, 300};
    size_t rows = 0; for (auto c : counts) rows += c;
    std::vector<T> v(rows);
    std::mt19937 g(5); std::uniform_real_distribution<double> u(0.5, 90.0);
    for (auto& x : v) x = (T)u(g);
    T *dv, *dm, *ds, *dp; uint64_t* dc;
    const uint64_t nseg = counts.size(), sc = segment_stats_scratch_values(nseg, rows);
    hipMalloc(&dv, rows * sizeof(T)); hipMalloc(&dm, nseg * sizeof(T)); hipMalloc(&ds, nseg * sizeof(T)); hipMalloc(&dp, sc * sizeof(T)); hipMalloc(&dc, nseg * 8);
    hipMemcpy(dv, v.data(), rows * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(dc, counts.data(), nseg * 8, hipMemcpyHostToDevice);
    hipMemset(dm, 0xFF, nseg * sizeof(T)); hipMemset(ds, 0xFF, nseg * sizeof(T));
    launch_segment_stats<T>(nullptr, dv, dc, 0, nseg, dp, dm, ds);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); return 1; }
    e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: launch: %s\n", name, hipGetErrorString(e)); return 1; }
    std::vector<T> hm(nseg), hs(nseg);
    hipMemcpy(hm.data(), dm, nseg * sizeof(T), hipMemcpyDeviceToHost); hipMemcpy(hs.data(), ds, nseg * sizeof(T), hipMemcpyDeviceToHost);
    size_t start = 0; int bad = 0;
    for (size_t s = 0; s < nseg; ++s) {
        const size_t n = counts[s];
        T mean = 0, sd = 0;
        if (n) {
            mean = np_sum(v.data() + start, n) / (T)n;
            std::vector<T> sq(n);
            for (size_t i = 0; i < n; ++i) { const T x = v[start + i] - mean; sq[i] = x * x; }
            sd = std::sqrt(np_sum(sq.data(), n) / (T)n);
        }
        if (std::memcmp(&mean, &hm[s], sizeof(T)) || std::memcmp(&sd, &hs[s], sizeof(T))) {
            printf("%s: n = %zu: mean %.17g vs %.17g, std %.17g vs %.17g\n", name, n, (double)hm[s], (double)mean, (double)hs[s], (double)sd);
            ++bad;
        }
        start += n;
    }
    printf("%s: %d of %zu segments differ\n", name, bad, (size_t)nseg);
    return bad;
}
int main() { return run<float>("float") + run<double>("double"); }

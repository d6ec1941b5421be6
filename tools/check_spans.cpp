// Host check of csrc/isect_spans.hpp: for random and adversarial Gaussians, the 16-byte span record the counting pass of the
// Gaussian-major intersection keeps per row (SpanPacker over walk_spans) must replay (span_tiles_visit) as exactly the tile
// sequence of walk_tiles, in order, whenever it says it fits; rows that do not fit (kSpanWalk) walk again in the emission.
// Also reports how often that happens. Build + run (host code only; no GPU needed):
//   hipcc -O2 -std=c++17 -ffp-contract=off -x hip --cuda-host-only -o /tmp/check_spans tools/check_spans.cpp && /tmp/check_spans
#include "../gsplat_amd/csrc/isect_spans.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace gsx;
namespace gsx { void set_last_error(const char *, ...) {} int check_launch(const char *) { return 0; } }

int main(int argc, char **argv)
{
    const long n_cases = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(4321);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    long bad = 0, total_tiles = 0, nonempty = 0, rewalk = 0, rewalk_small = 0, small = 0;
    const uint32_t tss[] = {16, 16, 8, 4, 12, 7, 16, 16};
    for (long it = 0; it < n_cases; ++it) {
        const int cfg = (int)(it % 8);
        const uint32_t ts = tss[cfg];
        // cfg 6: an 8K image (480 x 270 tiles: spans start beyond 255); cfg 7: a grid wider than the record's 12-bit fields
        const uint32_t W = cfg == 4 ? 1000 : cfg == 6 ? 7680 : cfg == 7 ? 70000 : 1920, H = cfg == 4 ? 700 : cfg == 6 ? 4320 : 1080;
        const uint32_t tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
        float mx = (U(rng) * 1.4f - 0.2f) * W, my = (U(rng) * 1.4f - 0.2f) * H;
        if (it % 17 == 0) mx = floorf(mx / ts) * ts; // centres on tile lines
        if (it % 19 == 0) my = floorf(my / ts) * ts + (it % 3 ? 0.f : 1e-4f);
        // sigma 0.37 .. 400 px; every third case is a typical splat of a trained scene (sigma below 12 px)
        const float sc = it % 3 == 0 ? expf(U(rng) * 3.5f - 1.f) : expf(U(rng) * 7.f - 1.f);
        const float s1 = sc, s2 = sc * expf(-U(rng) * 3.f);
        const float an = U(rng) * 6.2831853f, c = cosf(an), s = sinf(an);
        const float a = c * c * s1 * s1 + s * s * s2 * s2, b = c * s * (s1 * s1 - s2 * s2), d = s * s * s1 * s1 + c * c * s2 * s2;
        const float det = a * d - b * b;
        float A = d / det, B = -b / det, C = a / det;
        const float op = it % 11 == 0 ? U(rng) * 0.01f : U(rng);
        if (it % 101 == 0) B = 0.f;
        const bool has_conic = (it % 5) != 0;
        const float rx = ceilf(3.33f * sqrtf(a)), ry = ceilf(3.33f * sqrtf(d));
        std::vector<int64_t> ref, got;
        walk_tiles(mx, my, rx, ry, has_conic, A, B, C, op, ts, tw, th, [&](int64_t t) { ref.push_back(t); });
        SpanPacker sp;
        long n_in_spans = 0;
        walk_spans(mx, my, rx, ry, has_conic, A, B, C, op, ts, tw, th, [&](bool alongY, int u, int tv0, int tv1) {
            sp.add(alongY, u, tv0, tv1);
            n_in_spans += tv1 > tv0 ? tv1 - tv0 : 0;
        });
        if (n_in_spans != (long)ref.size()) ++bad; // the spans ARE the walk
        const SpanRecord rec = sp.record();
        const bool is_small = sc < 12.f && cfg < 4;
        small += is_small;
        if (span_slabs(rec) == kSpanWalk) {
            ++rewalk;
            rewalk_small += is_small;
            if (span_tiles(rec) < (1 << 20)) ++bad; // sorted in front of every recorded row
        } else {
            span_tiles_visit(rec, tw, [&](int64_t t) { got.push_back(t); });
            if (got != ref) {
                if (++bad < 10) fprintf(stderr, "mismatch case %ld cfg %d: ref %zu got %zu\n", it, cfg, ref.size(), got.size());
            }
            if (span_tiles(rec) != (int)ref.size()) ++bad;
            if (ref.empty() != (span_slabs(rec) == 0 || span_tiles(rec) == 0)) ++bad;
        }
        total_tiles += (long)ref.size();
        nonempty += !ref.empty();
    }
    printf("cases %ld nonempty %ld tiles %ld bad %ld; records that do not fit: %ld of all (%.2f %%), %ld of the %ld splats below 12 px sigma on a 1080p grid (%.3f %%)\n",
           n_cases, nonempty, total_tiles, bad, rewalk, 100.0 * rewalk / n_cases, rewalk_small, small, small ? 100.0 * rewalk_small / small : 0.0);
    return bad ? 1 : 0;
}

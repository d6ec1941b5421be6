// Host check of csrc/isect_binwalk.hpp: for random and adversarial Gaussians, the union of walk_clipped() over a
// partition of the tile grid into bins must be walk_tiles(), tile for tile (and every tile must come from the bin that
// contains it); so must the block records of the binned path (block_mask + block_bin_mask). Build + run: tools/check_binwalk.sh  (hipcc, host code only; no GPU needed)
#include "../gsplat_amd/csrc/isect_binwalk.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
using namespace gsx;
namespace gsx { void set_last_error(const char *, ...) {} int check_launch(const char *) { return 0; } }

int main(int argc, char **argv)
{
    const long n_cases = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    long bad = 0, total_tiles = 0, nonempty = 0;
    const uint32_t tss[] = {16, 16, 16, 8, 4, 12, 7, 16};
    const int bws[] = {4, 4, 2, 4, 4, 4, 3, 8}, bhs[] = {4, 2, 2, 4, 4, 4, 5, 2};
    for (long it = 0; it < n_cases; ++it) {
        const int cfg = (int)(it % 8);
        const uint32_t ts = tss[cfg];
        const int BW = bws[cfg], BH = bhs[cfg];
        const uint32_t W = cfg == 5 ? 1000 : 1920, H = cfg == 5 ? 700 : 1080;
        const uint32_t tw = (W + ts - 1) / ts, th = (H + ts - 1) / ts;
        // gaussian
        float mx = (U(rng) * 1.4f - 0.2f) * W, my = (U(rng) * 1.4f - 0.2f) * H;
        if (it % 17 == 0) { mx = floorf(mx / ts) * ts; }            // centres on tile lines
        if (it % 19 == 0) { my = floorf(my / ts) * ts + (it % 3 ? 0.f : 1e-4f); }
        const float sc = expf(U(rng) * 7.f - 1.f); // sigma 0.37 .. 400 px
        const float s1 = sc, s2 = sc * expf(-U(rng) * 3.f);
        const float th_ = U(rng) * 6.2831853f, c = cosf(th_), s = sinf(th_);
        // covariance = R diag(s1^2, s2^2) R^T ; conic = inverse
        const float a = c * c * s1 * s1 + s * s * s2 * s2, b = c * s * (s1 * s1 - s2 * s2), d = s * s * s1 * s1 + c * c * s2 * s2;
        const float det = a * d - b * b;
        float A = d / det, B = -b / det, C = a / det;
        float op = it % 11 == 0 ? U(rng) * 0.01f : U(rng);
        if (it % 101 == 0) { B = 0.f; }
        const bool has_conic = (it % 5) != 0;
        const float rx = ceilf(3.33f * sqrtf(a)), ry = ceilf(3.33f * sqrtf(d));
        std::vector<int64_t> ref, got, got_blocks;
        walk_tiles(mx, my, rx, ry, has_conic, A, B, C, op, ts, tw, th, [&](int64_t t) { ref.push_back(t); });
        const WalkPrep p = walk_prepare(mx, my, rx, ry, has_conic, A, B, C, op, ts, tw, th);
        if (p.any) {
            const int bx0 = p.x0 / BW, bx1 = (p.x1 + BW - 1) / BW, by0 = p.y0 / BH, by1 = (p.y1 + BH - 1) / BH;
            for (int by = by0; by < by1; ++by)
                for (int bx = bx0; bx < bx1; ++bx) {
                    const int cx0 = bx * BW, cy0 = by * BH, cx1 = std::min<int>(cx0 + BW, tw), cy1 = std::min<int>(cy0 + BH, th);
                    walk_clipped(p, ts, cx0, cy0, cx1, cy1, [&](int x, int y) {
                        if (x < cx0 || x >= cx1 || y < cy0 || y >= cy1) ++bad;
                        got.push_back((int64_t)y * tw + x);
                    });
                }
            // the block records of isect_binned.hip: the row's rectangle of bins cut into blocks of kx x ky bins, every block's
            // walk as one 64-bit word, every bin's 16-bit mask cut out of it - their union must be the walk as well
            if (BW * BH <= 16) {
                const int kx = 2, ky = BW * BH <= 8 ? 4 : 2, Wb = kx * BW, Hb = ky * BH;
                uint64_t col = 0;
                for (int y = 0; y < Hb; ++y) col |= 1ull << (y * Wb);
                for (int by = by0; by < by1; by += ky)
                    for (int bx = bx0; bx < bx1; bx += kx) {
                        const int cx0 = bx * BW, cy0 = by * BH;
                        const uint64_t M = block_mask(p, ts, tw, cx0, cy0, cx0 + Wb, cy0 + Hb, (uint32_t)Wb, col, nullptr);
                        uint64_t seen = 0;
                        for (int j = 0; j < ky; ++j)
                            for (int i = 0; i < kx; ++i) {
                                const uint32_t m = block_bin_mask(M, BW, BH, kx, i, j);
                                for (int b = 0; b < BW * BH; ++b)
                                    if ((m >> b) & 1u) {
                                        const int x = cx0 + i * BW + b % BW, y = cy0 + j * BH + b / BW;
                                        got_blocks.push_back((int64_t)y * tw + x);
                                        seen |= 1ull << ((y - cy0) * Wb + (x - cx0));
                                    }
                            }
                        if (seen != M) ++bad;
                    }
            }
            // and the unclipped call must reproduce walk_tiles in ORDER
            std::vector<int64_t> full;
            walk_clipped(p, ts, 0, 0, (int)tw, (int)th, [&](int x, int y) { full.push_back((int64_t)y * tw + x); });
            if (full != ref) ++bad;
        }
        for (int64_t t : ref) {
            const int x = (int)(t % tw), y = (int)(t / tw);
            if (!p.any || x < p.x0 || x >= p.x1 || y < p.y0 || y >= p.y1) ++bad; // the rectangle bounds the walk
        }
        std::sort(ref.begin(), ref.end());
        std::sort(got.begin(), got.end());
        std::sort(got_blocks.begin(), got_blocks.end());
        if (BW * BH <= 16 && ref != got_blocks) {
            if (++bad < 10) fprintf(stderr, "block mismatch case %ld cfg %d: ref %zu got %zu\n", it, cfg, ref.size(), got_blocks.size());
        }
        if (ref != got) {
            if (++bad < 10) fprintf(stderr, "mismatch case %ld cfg %d: ref %zu got %zu\n", it, cfg, ref.size(), got.size());
        }
        total_tiles += (long)ref.size();
        nonempty += !ref.empty();
    }
    printf("cases %ld nonempty %ld tiles %ld bad %ld\n", n_cases, nonempty, total_tiles, bad);
    return bad ? 1 : 0;
}

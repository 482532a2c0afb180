// Division of a compile-time base graph among the waves of a workgroup (multi-wave flooding kernels).
#pragma once
#include "spec_common.h"

namespace ldpc {

// Which wave owns which block row / block column: longest-first greedy balancing of the edge counts
// (plus one unit of fixed cost per row or column), evaluated at compile time.  Each phase ends in a
// barrier, so an iteration costs the most loaded wave's share.
template <int W, int N>
struct R1wAssign {
    int cnt[W];
    int item[W][N];
    int edge0[W][N + 1];  // running edge count in front of the wave's j-th item
    int max_cnt, max_edges;
};

template <int W, int N>
constexpr R1wAssign<W, N> r1w_balance(const int *ptr)
{
    R1wAssign<W, N> a{};
    int load[W] = {};
    bool used[N] = {};
    for (int step = 0; step < N; ++step) {
        int best = -1, bw = -1;
        for (int i = 0; i < N; ++i) {
            const int wgt = ptr[i + 1] - ptr[i];
            if (!used[i] && wgt > bw) { best = i; bw = wgt; }
        }
        used[best] = true;
        int w = 0;
        for (int v = 1; v < W; ++v)
            if (load[v] < load[w]) w = v;
        load[w] += bw + 1;
        a.item[w][a.cnt[w]++] = best;
    }
    for (int w = 0; w < W; ++w) {
        for (int i = 1; i < a.cnt[w]; ++i)  // ascending inside a wave
            for (int j = i; j > 0 && a.item[w][j - 1] > a.item[w][j]; --j) {
                const int t = a.item[w][j]; a.item[w][j] = a.item[w][j - 1]; a.item[w][j - 1] = t;
            }
        int e = 0;
        for (int j = 0; j < a.cnt[w]; ++j) {
            a.edge0[w][j] = e;
            e += ptr[a.item[w][j] + 1] - ptr[a.item[w][j]];
        }
        a.edge0[w][a.cnt[w]] = e;
        a.max_cnt = a.cnt[w] > a.max_cnt ? a.cnt[w] : a.max_cnt;
        a.max_edges = e > a.max_edges ? e : a.max_edges;
    }
    return a;
}

template <class C, int W>
struct R1wPlan {
    static constexpr R1wAssign<W, C::NB> COLS = r1w_balance<W, C::NB>(C::col_ptr);
    static constexpr R1wAssign<W, C::MB> ROWS = r1w_balance<W, C::MB>(C::row_ptr);
    static constexpr int NBW = COLS.max_cnt;                           // columns of the busiest wave
    static constexpr int QW = COLS.max_edges > 0 ? COLS.max_edges : 1;  // its edges
    static constexpr int MBW = ROWS.max_cnt;
};

}  // namespace ldpc

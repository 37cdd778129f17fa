// lsa.cpp — rectangular linear sum assignment on the host.
//
// Replaces scipy.optimize.linear_sum_assignment at
// /root/reference/mass/utils/experimentation.py:284-287.  scipy is a
// third-party dependency of the reference (unpinned; scipy 1.15.3 in the build
// image) whose solver is the shortest-augmenting-path method of
//   D. F. Crouse, "On implementing 2D rectangular assignment algorithms",
//   IEEE Trans. Aerospace and Electronic Systems 52(4), 2016.
// This file restates that published algorithm, including the two details that
// decide which optimum is returned when several exist (ties): the unvisited
// column list starts in descending column order and is compacted by moving its
// last element into the freed slot, and among equally cheap columns one that
// is still unassigned is preferred.  Instance counts are tiny (<= ~200), so it
// runs on the host like the reference's.
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>
#include "common.h"

namespace {

// One Dijkstra-like search from row `cur` over the reduced costs; returns the
// unassigned column reached (sink) or -1 if none is reachable.
int augmenting_path(int nc, const double *cost, const std::vector<double> &u, const std::vector<double> &v,
                    std::vector<int> &path, const std::vector<int> &row4col, std::vector<double> &dist, int cur,
                    std::vector<char> &SR, std::vector<char> &SC, std::vector<int> &remaining, double &min_val)
{
    const double INF = std::numeric_limits<double>::infinity();
    min_val = 0.0;
    int n_rem = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(dist.begin(), dist.end(), INF);
    int sink = -1, i = cur;
    while (sink == -1) {
        int index = -1;
        double lowest = INF;
        SR[i] = 1;
        for (int it = 0; it < n_rem; ++it) {
            const int j = remaining[it];
            const double r = min_val + cost[(size_t)i * nc + j] - u[i] - v[j];
            if (r < dist[j]) { path[j] = i; dist[j] = r; }
            if (dist[j] < lowest || (dist[j] == lowest && row4col[j] == -1)) { lowest = dist[j]; index = it; }
        }
        min_val = lowest;
        if (min_val == INF) return -1;
        const int j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--n_rem];
    }
    return sink;
}

}  // namespace

extern "C" int mf_linear_sum_assignment(const double *cost_in, int32_t n0, int32_t n1, int64_t *row_ind,
                                        int64_t *col_ind)
{
    if (n0 < 0 || n1 < 0) return mf::fail(MF_ERR_INVALID, "negative matrix dimension");
    if (n0 == 0 || n1 == 0) return 0;
    if (!cost_in || !row_ind || !col_ind) return mf::fail(MF_ERR_INVALID, "NULL pointer");
    for (size_t i = 0; i < (size_t)n0 * n1; ++i)
        if (std::isnan(cost_in[i]) || cost_in[i] == -std::numeric_limits<double>::infinity())
            return mf::fail(MF_ERR_INVALID, "matrix contains invalid numeric entries");

    // work on the orientation with rows <= columns
    const bool transposed = n1 < n0;
    int nr = n0, nc = n1;
    std::vector<double> tmp;
    const double *cost = cost_in;
    if (transposed) {
        tmp.resize((size_t)n0 * n1);
        for (int i = 0; i < n0; ++i)
            for (int j = 0; j < n1; ++j) tmp[(size_t)j * n0 + i] = cost_in[(size_t)i * n1 + j];
        cost = tmp.data();
        std::swap(nr, nc);
    }

    std::vector<double> u(nr, 0.0), v(nc, 0.0), dist(nc);
    std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);

    for (int cur = 0; cur < nr; ++cur) {
        double min_val;
        const int sink = augmenting_path(nc, cost, u, v, path, row4col, dist, cur, SR, SC, remaining, min_val);
        if (sink < 0) return mf::fail(MF_ERR_INVALID, "cost matrix is infeasible");
        // dual update
        u[cur] += min_val;
        for (int i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - dist[col4row[i]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - dist[j];
        // flip the alternating path back to `cur`
        int j = sink;
        for (;;) {
            const int i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }

    if (transposed) {
        std::vector<int> order(nr);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
        for (int i = 0; i < nr; ++i) { row_ind[i] = col4row[order[i]]; col_ind[i] = order[i]; }
    } else {
        for (int i = 0; i < nr; ++i) { row_ind[i] = i; col_ind[i] = col4row[i]; }
    }
    return nr;
}

// Host-side rectangular linear sum assignment (shortest augmenting paths, Crouse 2016), the
// algorithm behind the `scipy.optimize.linear_sum_assignment` the reference calls on the host at
// gom_lstmatcher.py:447 and :549.  The track ids depend on WHICH optimum is returned when several
// exist, so the tie-breaking follows SciPy's published behaviour: unassigned columns are scanned
// from the highest index down, a new sink wins ties, tall matrices are solved transposed.
// tests/test_lsa.py cross-checks >10^4 random and tied matrices against the installed SciPy.
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/gomatching_hip.h"

namespace {

long augmenting_path(long nc, const double* cost, std::vector<double>& u, std::vector<double>& v,
                     std::vector<long>& path, std::vector<long>& row4col, std::vector<double>& spc, long i,
                     std::vector<char>& SR, std::vector<char>& SC, std::vector<long>& remaining, double* p_min) {
    double min_val = 0;
    long num_remaining = nc;
    for (long it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), std::numeric_limits<double>::infinity());
    long sink = -1;
    while (sink == -1) {
        long index = -1;
        double lowest = std::numeric_limits<double>::infinity();
        SR[i] = 1;
        for (long it = 0; it < num_remaining; ++it) {
            const long j = remaining[it];
            const double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) {
                path[j] = i;
                spc[j] = r;
            }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == std::numeric_limits<double>::infinity()) return -1;
        const long j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

}  // namespace

extern "C" int gom_linear_sum_assignment(const double* cost_in, long nr, long nc, long* row_ind, long* col_ind) {
    if (nr < 0 || nc < 0 || !row_ind || !col_ind) return -GOM_ERR_INVALID_ARG;
    if (nr == 0 || nc == 0) return 0;
    if (!cost_in) return -GOM_ERR_INVALID_ARG;
    const bool transpose = nc < nr;
    std::vector<double> cost((size_t)nr * nc);
    if (transpose) {
        for (long i = 0; i < nr; ++i)
            for (long j = 0; j < nc; ++j) cost[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
    } else {
        std::copy(cost_in, cost_in + (size_t)nr * nc, cost.begin());
    }
    for (double c : cost)
        if (std::isnan(c) || c == -std::numeric_limits<double>::infinity()) return -GOM_ERR_INVALID_ARG;

    std::vector<double> u(nr, 0), v(nc, 0), spc(nc);
    std::vector<long> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (long cur = 0; cur < nr; ++cur) {
        double min_val;
        const long sink = augmenting_path(nc, cost.data(), u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) return -GOM_ERR_UNSUPPORTED;           // infeasible
        u[cur] += min_val;
        for (long i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (long j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        long j = sink;
        while (true) {
            const long i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) {
        std::vector<long> order(nr);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](long a, long b) { return col4row[a] < col4row[b]; });
        for (long i = 0; i < nr; ++i) {
            row_ind[i] = col4row[order[i]];
            col_ind[i] = order[i];
        }
    } else {
        for (long i = 0; i < nr; ++i) {
            row_ind[i] = i;
            col_ind[i] = col4row[i];
        }
    }
    return (int)nr;
}

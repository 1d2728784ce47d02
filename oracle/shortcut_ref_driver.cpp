// oracle/shortcut_ref_driver.cpp -- TEST INFRASTRUCTURE.  A driver (this file is ours) around the REFERENCE's own
// header-only, std-only shortcutting routine, compiled from where it lies: -I/root/reference/smpl/include
// (smpl/include/smpl/geometry/shortcut.h + detail/shortcut.hpp), the loop behind smpl::ShortcutPath
// (smpl/src/post_processing.cpp:281-309).  Output only goes to oracle/_ref/.  It generates
// tests/golden/shortcut_ref.json through tests/golden/make_shortcut_golden.py; it never travels as source.
//
// Points are indices 0..P-1.  The one path generator answers from tables, like JointPositionShortcutPathGenerator
// (post_processing.cpp:100-127) answers from the collision checker: valid[i][j] -> path {i, j} with cost[i][j].
// stdin: P, then P*P costs, then P*P validity flags (row-major).  stdout: the shortcut path's indices.
#include <smpl/geometry/shortcut.h>

#include <cstdio>
#include <vector>

static int g_P = 0;
static std::vector<double> g_cost;
static std::vector<int> g_valid;

struct TableGenerator : public sbpl::shortcut::PathGenerator<int, double>
{
    bool generate_path(const int& start, const int& end, std::vector<int>& path_out, double& cost_out) const override
    {
        if (!g_valid[(size_t)start * g_P + end]) return false;
        path_out = {start, end};
        cost_out = g_cost[(size_t)start * g_P + end];
        return true;
    }
};

int main()
{
    if (scanf("%d", &g_P) != 1 || g_P < 0) return 1;
    g_cost.resize((size_t)g_P * g_P);
    g_valid.resize((size_t)g_P * g_P);
    for (double& c : g_cost) if (scanf("%lf", &c) != 1) return 1;
    for (int& v : g_valid) if (scanf("%d", &v) != 1) return 1;
    std::vector<int> points(g_P), out;
    for (int i = 0; i < g_P; ++i) points[i] = i;
    std::vector<double> costs;
    for (int i = 0; i + 1 < g_P; ++i) costs.push_back(g_cost[(size_t)i * g_P + i + 1]);
    std::vector<TableGenerator> gens(1);
    const bool ok = sbpl::shortcut::ShortcutPath(points, costs, gens, out);   // window 1, granularity 1, std::less_equal
    printf("%d %zu", ok ? 1 : 0, out.size());
    for (int i : out) printf(" %d", i);
    printf("\n");
    return 0;
}

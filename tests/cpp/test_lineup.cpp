// CPU check of rustyhgi_amd/csrc/hgi_lineup.h (which chunk of which memory class goes where in a set of composed planes):
// thousands of synthetic classifications -- group counts, sizes and creation orders as the driver produces them (runs of one
// class) and as it does not (shuffled) -- against the promises the header states.  Built by the CPU suite with
// g++ -fsanitize=address,undefined (tests/test_sanitizers.py).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <array>
#include <set>

#include "../../rustyhgi_amd/csrc/hgi_lineup.h"

using namespace hgi::lineup;

#define REQUIRE(cond)                                                          \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::printf("FAILED %s:%d: %s (case %d)\n", __FILE__, __LINE__, #cond, g_case); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static int g_case = 0;

static int group_of(const Groups &groups, int chunk)
{
    for (size_t g = 0; g < groups.size(); ++g)
        for (int j : groups[g])
            if (j == chunk) return (int)g;
    return -1;
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(0x48474930);
    int sided = 0, offset_only = 0, neither = 0;
    for (g_case = 0; g_case < cases; ++g_case) {
        const size_t n = 1 + rng() % 16;                  // chunks per plane
        const uint32_t count = 2 + (uint32_t)(rng() % 4);      // planes
        const size_t created = n * count + rng() % (2 * n * count + 1);
        const size_t G = 1 + rng() % 6;
        // classes in runs, as the driver hands memory out, or shuffled
        std::vector<int> cls(created);
        if (rng() % 3) {
            size_t at = 0;
            int c = (int)(rng() % G);
            while (at < created) {
                const size_t run = 1 + rng() % (2 * n + 3);
                for (size_t k = 0; k < run && at < created; ++k) cls[at++] = c;
                c = (int)(rng() % G);
            }
        } else {
            for (auto &c : cls) c = (int)(rng() % G);
        }
        // groups in order of first appearance (what classify() builds)
        Groups groups;
        std::vector<int> label(G, -1);
        for (size_t j = 0; j < created; ++j) {
            if (label[(size_t)cls[j]] < 0) {
                label[(size_t)cls[j]] = (int)groups.size();
                groups.push_back({});
            }
            groups[(size_t)label[(size_t)cls[j]]].push_back((int)j);
        }
        Rows rows;
        size_t reported = 99;
        bool two = two_sides(groups, n, count, rows, &reported), complete = two;
        if (two) {
            ++sided;
            REQUIRE(count < 2 || odd_spread_of(groups, rows, count) == reported);
            // what alloc_composed does when the odd planes come out (almost) on one class: the per-offset line-up of the same
            // groups, if complete, keeps the promise of every offset and uses every chunk once
            Rows alt;
            if (alternating(groups, n, count, alt)) {
                REQUIRE(alt.size() == n);
                std::set<int> once;
                for (auto &row : alt) {
                    for (uint32_t i = 0; i + 1 < count; ++i) REQUIRE(group_of(groups, row[i]) != group_of(groups, row[i + 1]));
                    for (int j : row) REQUIRE(once.insert(j).second);
                }
                REQUIRE(odd_spread_of(groups, alt, count) <= n);
            }
            // every chunk of a plane differs in group from EVERY chunk of its neighbours
            for (uint32_t i = 0; i + 1 < count; ++i)
                for (size_t a = 0; a < n; ++a)
                    for (size_t b = 0; b < n; ++b) REQUIRE(group_of(groups, rows[a][i]) != group_of(groups, rows[b][i + 1]));
        } else {
            complete = per_offset(groups, n, count, rows);
            REQUIRE(rows.size() <= n);
            if (complete) ++offset_only; else ++neither;
            // a two-sided line-up exists whenever some split of the groups holds enough chunks on both sides: brute force says no
            const size_t need_even = n * ((count + 1) / 2), need_odd = n * (count / 2);
            for (size_t mask = 1; mask + 1 < ((size_t)1 << groups.size()); ++mask) {
                size_t x = 0, y = 0;
                for (size_t g = 0; g < groups.size(); ++g) ((mask >> g) & 1 ? x : y) += groups[g].size();
                REQUIRE(!(x >= need_even && y >= need_odd));
            }
            const size_t lined = rows.size();
            fill_rest(rows, n, count, created);
            // the offsets that lined up keep their promise; the rest is filled with unused chunks
            for (size_t m = 0; m < lined; ++m)
                for (uint32_t i = 0; i + 1 < count; ++i) REQUIRE(group_of(groups, rows[m][i]) != group_of(groups, rows[m][i + 1]));
        }
        REQUIRE(rows.size() == n);
        std::set<int> seen;
        for (auto &row : rows) {
            REQUIRE(row.size() == count);
            for (int j : row) {
                REQUIRE(j >= 0 && (size_t)j < created);
                REQUIRE(seen.insert(j).second);      // every chunk at most once
            }
        }
        if (complete)
            for (size_t m = 0; m < n; ++m)
                for (uint32_t i = 0; i + 1 < count; ++i) REQUIRE(group_of(groups, rows[m][i]) != group_of(groups, rows[m][i + 1]));
        // stalled(): true exactly when the last `look` chunks created are the last `look` members of the largest group
        const size_t look = n < 4 ? n : 4;
        size_t big = 0;
        const bool st = stalled(groups, created, look, &big);
        size_t b = 0;
        for (size_t g = 1; g < groups.size(); ++g)
            if (groups[g].size() > groups[b].size()) b = g;
        bool want = groups[b].size() >= look;
        for (size_t t = 0; t < look && want; ++t) want = label[(size_t)cls[created - 1 - t]] == (int)b;
        REQUIRE(st == want);
    }
    // the shapes seen on the device (profiles/r04_planes_sides.txt): 512 frames = 8 chunks x 3 planes
    {
        Groups g = {{}, {}, {}};
        int j = 0;
        for (int k = 0; k < 3; ++k) g[0].push_back(j++);
        for (int k = 0; k < 6; ++k) g[1].push_back(j++);
        for (int k = 0; k < 19; ++k) g[2].push_back(j++);
        Rows rows;
        REQUIRE(two_sides(g, 8, 3, rows));
        int from0 = 0, from1 = 0;
        for (size_t m = 0; m < 8; ++m) {
            REQUIRE(group_of(g, rows[m][0]) == 2 && group_of(g, rows[m][2]) == 2 && group_of(g, rows[m][1]) != 2);
            (group_of(g, rows[m][1]) == 0 ? from0 : from1)++;
        }
        REQUIRE(from0 == 3 && from1 == 5);      // the grid plane is spread over both small classes as evenly as 3 + 6 chunks allow
        // 7 / 5 / 16: the grid plane takes 4 + 4, alternating
        Groups h7 = {{}, {}, {}};
        j = 0;
        for (int k = 0; k < 7; ++k) h7[0].push_back(j++);
        for (int k = 0; k < 5; ++k) h7[1].push_back(j++);
        for (int k = 0; k < 16; ++k) h7[2].push_back(j++);
        REQUIRE(two_sides(h7, 8, 3, rows));
        for (size_t m = 0; m + 1 < 8; ++m) REQUIRE(group_of(h7, rows[m][1]) != group_of(h7, rows[m + 1][1]) && group_of(h7, rows[m][1]) != 2);
        // 11 / 15 / 6 (a box of the pool): the grid plane cannot be spread, the image planes are (5 + 3); ONE more chunk of the second
        // class and the sides swap: the grid plane gets 4 + 4 -- which is why alloc_composed keeps looking a little longer
        {
            Groups b = {{}, {}, {}};
            int t = 0;
            for (int k = 0; k < 11; ++k) b[0].push_back(t++);
            for (int k = 0; k < 15; ++k) b[1].push_back(t++);
            for (int k = 0; k < 6; ++k) b[2].push_back(t++);
            size_t sp = 99;
            REQUIRE(two_sides(b, 8, 3, rows, &sp) && sp == 0);
            b[1].push_back(t++);
            REQUIRE(two_sides(b, 8, 3, rows, &sp) && sp == 4);
            for (size_t m = 0; m < 8; ++m) REQUIRE(group_of(b, rows[m][0]) == 1 && group_of(b, rows[m][2]) == 1 && group_of(b, rows[m][1]) != 1);
        }
        // 8 / 8 / 8: two classes share the image planes half and half, the third is the grid
        Groups e8 = {{}, {}, {}};
        j = 0;
        for (int gi = 0; gi < 3; ++gi)
            for (int k = 0; k < 8; ++k) e8[(size_t)gi].push_back(j++);
        size_t spread = 99;
        REQUIRE(two_sides(e8, 8, 3, rows, &spread) && spread == 0);      // the one odd plane sits on one class: nothing else left
        for (int pl : {0, 2}) {
            int a = 0;
            for (size_t m = 0; m < 8; ++m) a += group_of(e8, rows[m][(size_t)pl]) == group_of(e8, rows[0][(size_t)pl]);
            REQUIRE(a == 4);
        }
        // two classes is all there is (18 / 18): two sides leave the grid plane on one class; the per-offset line-up alternates every
        // plane (4 + 4), which is what alloc_composed then takes (odd_spread_of tells it)
        {
            Groups two = {{}, {}};
            int t = 0;
            for (int k = 0; k < 18; ++k) two[0].push_back(t++);
            for (int k = 0; k < 18; ++k) two[1].push_back(t++);
            size_t sp = 99;
            REQUIRE(two_sides(two, 8, 3, rows, &sp) && sp == 0 && odd_spread_of(two, rows, 3) == 0);
            Rows alt;
            REQUIRE(alternating(two, 8, 3, alt) && odd_spread_of(two, alt, 3) == 4);
            for (size_t m = 0; m < 8; ++m) {
                REQUIRE(group_of(two, alt[m][0]) != group_of(two, alt[m][1]) && group_of(two, alt[m][1]) != group_of(two, alt[m][2]));
                if (m) REQUIRE(group_of(two, alt[m][1]) != group_of(two, alt[m - 1][1]));
            }
            // uneven supplies, as a device hands them out: 27 / 13 still gives the grid plane 4 + 4, 28 / 10 gives it 6 + 2 (an
            // offset at which the grid plane takes the large class costs two chunks of the small one), 30 / 8 nothing
            for (auto want : {std::array<size_t, 3>{{27, 13, 4}}, std::array<size_t, 3>{{28, 10, 2}}, std::array<size_t, 3>{{30, 8, 0}}, std::array<size_t, 3>{{9, 11, 4}}}) {
                Groups u = {{}, {}};
                int v = 0;
                for (size_t k = 0; k < want[0]; ++k) u[0].push_back(v++);
                for (size_t k = 0; k < want[1]; ++k) u[1].push_back(v++);
                const size_t nn = want[0] == 9 ? 4 : 8;
                REQUIRE(alternating(u, nn, 3, alt) && alt.size() == nn);
                for (auto &row : alt) REQUIRE(group_of(u, row[0]) != group_of(u, row[1]) && group_of(u, row[1]) != group_of(u, row[2]));
                REQUIRE(odd_spread_of(u, alt, 3) == (want[0] == 9 ? 2 : want[2]));
            }
            REQUIRE(two_sides(h7, 8, 3, rows, &sp) && sp == 4 && odd_spread_of(h7, rows, 3) == 4);
            // 24 / 14 / 1 / 1 (a box of the pool): the two lone chunks are no classes to spread a grid plane over -- two sides report 0,
            // and the alternating line-up of the same groups gives 4 + 4
            Groups lone = {{}, {}, {38}, {39}};
            t = 0;
            for (int k = 0; k < 24; ++k) lone[0].push_back(t++);
            for (int k = 0; k < 14; ++k) lone[1].push_back(t++);
            REQUIRE(two_sides(lone, 8, 3, rows, &sp) && sp == 0);
            REQUIRE(alternating(lone, 8, 3, alt) && odd_spread_of(lone, alt, 3) >= 3);
        }
        Groups one = {{0, 1, 2, 3, 4, 5}};
        REQUIRE(!two_sides(one, 2, 3, rows) && !per_offset(one, 2, 3, rows));
    }
    std::printf("%d cases ok: %d two-sided, %d per offset only, %d not separable\n", cases, sided, offset_only, neither);
    return 0;
}

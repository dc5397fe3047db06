// Which classified chunk goes where in a set of composed planes (hgi_planes.hip, alloc_composed) -- plain C++, no HIP, so that
// the CPU suite can build it with g++ -fsanitize=address,undefined and check its promises on thousands of synthetic
// classifications (tests/cpp/test_lineup.cpp).
//
// Input: `groups` -- the chunks created so far, sorted into groups that share a memory class (indices into the creation order);
// `n` chunks per plane, `count` planes.  Output: rows[offset][plane] = chunk.  Promise of a COMPLETE line-up: every chunk is
// used at most once, and at every offset neighbouring planes sit on chunks of different groups.  A TWO-SIDED line-up promises
// more: every chunk of a plane differs in group from EVERY chunk of its neighbouring planes.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <utility>
#include <vector>

namespace hgi {
namespace lineup {

typedef std::vector<std::vector<int>> Groups;
typedef std::vector<std::vector<int>> Rows;

// Greedy arrangement of `count` members such that neighbours come from different groups: always take from the largest
// remaining group that is not the one just used.  `left[g]` = members of group g still available (updated on success only).
// Returns the group of every position, or an empty vector if it cannot be done.
inline std::vector<int> arrange(std::vector<size_t> &left, uint32_t count)
{
    std::vector<int> seq;
    std::vector<size_t> trial = left;
    int prev = -1;
    while (seq.size() < count) {
        int pick = -1;
        for (size_t g = 0; g < trial.size(); ++g)
            if ((int)g != prev && trial[g] > 0 && (pick < 0 || trial[g] > trial[(size_t)pick])) pick = (int)g;
        if (pick < 0) return {};
        seq.push_back(pick);
        --trial[(size_t)pick];
        prev = pick;
    }
    left = trial;
    return seq;
}

// First choice: TWO SIDES.  The groups are split into a side for the even planes (image, image') and a side for the odd ones
// (grid) such that each side holds enough chunks; among the feasible splits the one whose largest groups carry most of their
// side wins (planes as uniform in class as can be).  Planes of a side take consecutive runs of the side's members, larger groups
// first.  Returns false (rows untouched) when no split is feasible yet.
inline bool two_sides(const Groups &groups, size_t n, uint32_t count, Rows &rows)
{
    const size_t G = groups.size(), need_even = n * ((count + 1) / 2), need_odd = n * (count / 2);
    if (G < 2 || G > 16 || count < 2) return false;
    size_t best = 0, best_score = 0;
    for (size_t mask = 1; mask + 1 < ((size_t)1 << G); ++mask) {
        size_t x = 0, y = 0, xmax = 0, ymax = 0;
        for (size_t g = 0; g < G; ++g) {
            const size_t sz = groups[g].size();
            if ((mask >> g) & 1) {
                x += sz;
                xmax = sz > xmax ? sz : xmax;
            } else {
                y += sz;
                ymax = sz > ymax ? sz : ymax;
            }
        }
        const size_t score = (xmax < need_even ? xmax : need_even) + (ymax < need_odd ? ymax : need_odd);
        if (x >= need_even && y >= need_odd && score > best_score) {
            best = mask;
            best_score = score;
        }
    }
    if (!best) return false;
    std::vector<size_t> by_size(G);
    for (size_t g = 0; g < G; ++g) by_size[g] = g;
    for (size_t a = 0; a < G; ++a)
        for (size_t b = a + 1; b < G; ++b)
            if (groups[by_size[b]].size() > groups[by_size[a]].size()) std::swap(by_size[a], by_size[b]);
    std::vector<int> side[2];      // [0] even planes, [1] odd planes; members of the larger groups first
    for (size_t g : by_size)
        for (int j : groups[g]) side[(best >> g) & 1 ? 0 : 1].push_back(j);
    rows.clear();
    for (size_t m = 0; m < n; ++m) {
        std::vector<int> row;
        for (uint32_t i = 0; i < count; ++i) row.push_back(side[i & 1][(size_t)(i / 2) * n + m]);
        rows.push_back(row);
    }
    return true;
}

// Second choice, at the end of the budget: per offset -- neighbouring planes differ at every offset, but the sides may flip
// along the plane.  Fills rows as far as it gets; returns whether all n offsets lined up.
inline bool per_offset(const Groups &groups, size_t n, uint32_t count, Rows &rows)
{
    std::vector<size_t> left(groups.size()), next(groups.size(), 0);
    for (size_t g = 0; g < groups.size(); ++g) left[g] = groups[g].size();
    rows.clear();
    for (size_t m = 0; m < n; ++m) {
        const std::vector<int> seq = arrange(left, count);
        if (seq.empty()) return false;
        std::vector<int> row;
        for (int g : seq) row.push_back(groups[(size_t)g][next[(size_t)g]++]);
        rows.push_back(row);
    }
    return true;
}

// Could not be established within the budget: the offsets that did line up stay as they are (a partly separated stream is still
// faster), the others take what is left, in creation order.  `created` >= n * count chunks exist.
inline void fill_rest(Rows &rows, size_t n, uint32_t count, size_t created)
{
    std::vector<char> used(created, 0);
    for (auto &row : rows)
        for (int j : row) used[(size_t)j] = 1;
    size_t at = 0;
    while (rows.size() < n) {
        std::vector<int> row;
        while (row.size() < count) {
            while (used[at]) ++at;
            used[at] = 1;
            row.push_back((int)at);
        }
        rows.push_back(row);
    }
}

// The driver hands out physical memory in runs of one class: did the last `look` chunks created all join the largest group?
// (Then a spacer in front of the next ones takes the rest of the run away.)  `big` receives that group.
inline bool stalled(const Groups &groups, size_t created, size_t look, size_t *big)
{
    if (groups.empty() || created < look || look == 0) return false;
    size_t b = 0;
    for (size_t g = 1; g < groups.size(); ++g)
        if (groups[g].size() > groups[b].size()) b = g;
    *big = b;
    if (groups[b].size() < look) return false;
    for (size_t t = 0; t < look; ++t)
        if (groups[b][groups[b].size() - 1 - t] != (int)(created - 1 - t)) return false;
    return true;
}

}  // namespace lineup
}  // namespace hgi

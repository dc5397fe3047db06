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

// How many chunks each group of a side gives to ONE plane of that side: proportional shares of what the groups still hold
// (`left`, over the `planes_left` planes of the side still to be served), trimmed from the largest share / topped up where most
// is left until they add up to n.  Spreads a plane over the classes of its side as evenly as the supply allows.
inline std::vector<size_t> plane_quota(const std::vector<size_t> &left, size_t planes_left, size_t n)
{
    std::vector<size_t> q(left.size());
    size_t sum = 0;
    for (size_t g = 0; g < left.size(); ++g) sum += q[g] = left[g] / planes_left;
    while (sum > n) {
        size_t big = 0;
        for (size_t g = 1; g < q.size(); ++g)
            if (q[g] > q[big]) big = g;
        --q[big];
        --sum;
    }
    while (sum < n) {
        size_t pick = q.size();
        for (size_t g = 0; g < q.size(); ++g)
            if (left[g] > q[g] && (pick == q.size() || left[g] - q[g] > left[pick] - q[pick])) pick = g;
        if (pick == q.size()) break;      // (cannot happen while the side holds n * planes_left chunks)
        ++q[pick];
        ++sum;
    }
    return q;
}

// How far a plane is spread: of its n chunks, taken q[g] from group g, how many lie OUTSIDE the class it has most of.  A group with
// a single member does not count as a class of its own here: such chunks are typically the ones that straddle two classes (their
// probes come out between the populations against every group: 0.966 / 0.970 / 0.971 of the yardstick where members are at 0.98-1.03
// and strangers at 0.92-0.95) -- fine as neighbours, but a grid plane on 6 + 1 + 1 of them runs like one on 8 + 0
// (profiles/r04_two_classes.txt).
inline size_t outside_largest(const Groups &groups, const std::vector<size_t> &q, size_t n)
{
    size_t big = 0, lone = 0;
    for (size_t g = 0; g < q.size(); ++g) {
        if (groups[g].size() < 2)
            lone += q[g];
        else
            big = q[g] > big ? q[g] : big;
    }
    return n > big + lone ? n - big - lone : 0;
}

// First choice: TWO SIDES.  The groups are split into a side for the even planes (image, image') and a side for the odd ones
// (grid) such that each side holds enough chunks.  Every plane is then spread over the groups of its side as EVENLY as the
// supply allows (plane_quota), its chunks alternating between them: an encode of 4 GiB and more per plane is dealt to the XCDs as
// contiguous eighths, i.e. eight chunks of each plane are in use at one time, and the more classes those touch the faster it
// runs -- measured on 512 x 4096^2 with the grid plane on 8 + 0 / 7 + 1 / 6 + 2 / 4 + 4 chunks of two classes: 2.787 / 2.730 /
// 2.670 / 2.654 ms (profiles/r04_planes_sides.txt, r04_c3_xcd_boxes.txt); how the image planes are spread does not show.  Among the
// feasible splits the one with the most evenly spread planes wins, the odd planes (what an encode WRITES) counting twice.
// Returns false (rows untouched) when no split is feasible yet.  *odd_spread (optional) receives, for the chosen line-up, the
// smallest number of chunks any odd plane has OUTSIDE its largest class (outside_largest; 0: some grid plane sits on one class only).
inline bool two_sides(const Groups &groups, size_t n, uint32_t count, Rows &rows, size_t *odd_spread = nullptr)
{
    const size_t G = groups.size();
    const size_t planes_of[2] = {(size_t)(count + 1) / 2, (size_t)count / 2};      // even planes, odd planes
    if (G < 2 || G > 16 || count < 2) return false;
    // quotas[side][plane of the side][group] for a split; score = chunks that do NOT sit in their plane's largest share
    auto evaluate = [&](size_t mask, std::vector<std::vector<size_t>> (&quota)[2], size_t *spread) -> long {
        long score = 0;
        *spread = n;
        for (int sd = 0; sd < 2; ++sd) {
            std::vector<size_t> left(G, 0);
            size_t have = 0;
            for (size_t g = 0; g < G; ++g)
                if ((((mask >> g) & 1) != 0) == (sd == 0)) have += left[g] = groups[g].size();
            if (have < n * planes_of[sd]) return -1;
            quota[sd].clear();
            for (size_t p = 0; p < planes_of[sd]; ++p) {
                std::vector<size_t> q = plane_quota(left, planes_of[sd] - p, n);
                size_t sum = 0;
                for (size_t g = 0; g < G; ++g) {
                    sum += q[g];
                    left[g] -= q[g];
                }
                if (sum != n) return -1;
                const size_t out = outside_largest(groups, q, n);
                score += (long)(sd == 1 ? 2 : 1) * (long)out;
                if (sd == 1 && out < *spread) *spread = out;
                quota[sd].push_back(q);
            }
        }
        return score;
    };
    size_t best = 0, best_spread = 0, spread = 0;
    long best_score = -1;
    std::vector<std::vector<size_t>> quota[2], best_quota[2];
    for (size_t mask = 1; mask + 1 < ((size_t)1 << G); ++mask) {
        const long score = evaluate(mask, quota, &spread);
        if (score > best_score) {
            best = mask;
            best_score = score;
            best_spread = spread;
            best_quota[0] = quota[0];
            best_quota[1] = quota[1];
        }
    }
    if (!best) return false;
    if (odd_spread) *odd_spread = best_spread;
    // hand the chunks out: plane by plane, the plane's groups taking turns (largest remaining share first)
    std::vector<size_t> next(G, 0);
    Rows plane_chunks(count);
    for (uint32_t i = 0; i < count; ++i) {
        std::vector<size_t> q = best_quota[i & 1][i / 2];
        int prev = -1;
        while (plane_chunks[i].size() < n) {
            int pick = -1;
            for (size_t g = 0; g < G; ++g)
                if (q[g] > 0 && (int)g != prev && (pick < 0 || q[g] > q[(size_t)pick])) pick = (int)g;
            if (pick < 0) pick = prev;      // only one group left: no more turns to take
            plane_chunks[i].push_back(groups[(size_t)pick][next[(size_t)pick]++]);
            --q[(size_t)pick];
            prev = pick;
        }
    }
    rows.clear();
    for (size_t m = 0; m < n; ++m) {
        std::vector<int> row;
        for (uint32_t i = 0; i < count; ++i) row.push_back(plane_chunks[i][m]);
        rows.push_back(row);
    }
    return true;
}

// Last choice, at the end of the budget without two sides: per offset -- neighbouring planes differ at every offset, but the sides
// may flip along the plane.  Size-greedy (whatever completes).  Fills rows as far as it gets; returns whether all n offsets lined up.
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

// Second choice, where two sides leave a grid plane on one class (two classes are all the search found): per offset again, but
// SPREADING every plane over the groups: at each offset a plane prefers the group it has used least so far
// (ties: the group with most left), as long as the remaining offsets can still be completed the size-greedy way; where they could
// not, the offset is arranged size-greedy itself.  With two groups of 12 and more chunks each, three planes of eight alternate
// perfectly (4 + 4 each); with 27 / 13 the grid plane still gets 4 + 4; with 28 / 10, 6 + 2.  Returns whether all n offsets
// lined up (the same promise as per_offset).
inline bool alternating(const Groups &groups, size_t n, uint32_t count, Rows &rows)
{
    const size_t G = groups.size();
    std::vector<size_t> left(G), next(G, 0);
    for (size_t g = 0; g < G; ++g) left[g] = groups[g].size();
    std::vector<std::vector<size_t>> used(count, std::vector<size_t>(G, 0));      // [plane][group] chunks taken so far
    auto can_finish = [&](std::vector<size_t> trial, size_t offsets) {
        for (size_t m = 0; m < offsets; ++m)
            if (arrange(trial, count).empty()) return false;
        return true;
    };
    rows.clear();
    for (size_t m = 0; m < n; ++m) {
        std::vector<size_t> trial = left;
        std::vector<int> seq;
        int prev = -1;
        for (uint32_t i = 0; i < count; ++i) {
            int pick = -1;
            for (size_t g = 0; g < G; ++g) {
                if ((int)g == prev || trial[g] == 0) continue;
                if (pick < 0 || used[i][g] < used[i][(size_t)pick] || (used[i][g] == used[i][(size_t)pick] && trial[g] > trial[(size_t)pick])) pick = (int)g;
            }
            if (pick < 0) break;
            seq.push_back(pick);
            --trial[(size_t)pick];
            prev = pick;
        }
        if (seq.size() < count || !can_finish(trial, n - m - 1)) {
            trial = left;
            seq = arrange(trial, count);
            if (seq.empty()) return false;
        }
        left = trial;
        std::vector<int> row;
        for (uint32_t i = 0; i < count; ++i) {
            const size_t g = (size_t)seq[i];
            ++used[i][g];
            row.push_back(groups[g][next[g]++]);
        }
        rows.push_back(row);
    }
    return true;
}

// For a line-up `rows` over `groups`: the smallest number of chunks any odd plane has OUTSIDE its largest class (what two_sides
// reports as *odd_spread).
inline size_t odd_spread_of(const Groups &groups, const Rows &rows, uint32_t count)
{
    size_t chunks = 0;
    for (auto &g : groups)
        for (int j : g) chunks = (size_t)j + 1 > chunks ? (size_t)j + 1 : chunks;
    std::vector<int> group_of(chunks, -1);
    for (size_t g = 0; g < groups.size(); ++g)
        for (int j : groups[g]) group_of[(size_t)j] = (int)g;
    size_t spread = rows.size();
    for (uint32_t i = 1; i < count; i += 2) {
        std::vector<size_t> per(groups.size(), 0);
        for (auto &row : rows) ++per[(size_t)group_of[(size_t)row[i]]];
        const size_t out = outside_largest(groups, per, rows.size());
        if (out < spread) spread = out;
    }
    return spread;
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

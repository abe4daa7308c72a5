// pair_order.h -- processing order of a pair list: launch batches that touch few stations (host code).
//
// A launch batch of the pair pass reads, per column k2, the 12 KB spectrum column of every station
// that occurs in the batch; the first touch of a (station, k2) column inside a launch is what crosses
// the L2 <-> fabric boundary (consecutive launches evict each other: ~14 MB of T and CCF traffic per
// XCD and launch against 4 MiB of L2).  So the fabric read traffic of the spectra is
//   (distinct stations of the batch) x (spectrum bytes) / (pair-windows of the batch),
// and the order of the pair list decides it:  25 consecutive pairs of an 8 x 8 station tile touch 12
// stations (0.48 per pair), a 5 x 5 biclique 10 (0.40), the 28 pairs among 8 stations 8 (0.29).
//
// pair_groups() cuts a pair list into groups of at most Bmax pairs:
//   1. S <= 64: the 72 lines of the affine plane AG(2, 8) (station s = point (s >> 3, s & 7) over
//      GF(8)) partition ALL pairs of 64 points into 8-cliques; every line that still holds at least
//      3/4 Bmax unassigned pairs (and at most Bmax) becomes a group.  For the all-pairs list of 64
//      stations that is the whole job: 72 groups of 28 pairs on 8 stations.
//   2. whatever is left: greedy station sets -- start from the station with the most unassigned
//      pairs, keep adding the station that brings the most unassigned pairs into the set while the
//      gain is at least half the set's pairs-per-station and the group stays within Bmax.  Cliques,
//      bicliques and anything between fall out of the same rule.
//   3. small leftovers are merged up to Bmax.
// perm = processing order (indices into the pair list), bounds = group starts, bounds.back() = P.
#pragma once
#include <algorithm>
#include <vector>

namespace xcx {

inline int gf8_mul(int a, int b) {
  int r = 0;
  for (int i = 0; i < 3; ++i)
    if (b & (1 << i)) r ^= a << i;
  for (int i = 4; i >= 3; --i)
    if (r & (1 << i)) r ^= 0xB << (i - 3);          // x^3 + x + 1
  return r & 7;
}

inline void pair_groups(const int* pairs, int S, int P, int Bmax, std::vector<int>& perm, std::vector<int>& bounds) {
  perm.clear(); bounds.clear();
  perm.reserve(P);
  if (Bmax < 1) Bmax = 1;
  // adjacency: per station the (other station, pair index) of every pair; self pairs appear once
  std::vector<std::vector<std::pair<int, int>>> adj(S);
  for (int q = 0; q < P; ++q) {
    const int a = pairs[2 * q], b = pairs[2 * q + 1];
    adj[a].push_back({b, q});
    if (b != a) adj[b].push_back({a, q});
  }
  std::vector<char> used(P, 0);
  std::vector<int> deg(S);
  for (int s = 0; s < S; ++s) deg[s] = (int)adj[s].size();
  std::vector<std::vector<int>> groups;
  std::vector<char> in_set(S, 0);
  // unassigned pairs with both stations in the marked set, in (first station, list) order
  auto collect = [&](const std::vector<int>& set, std::vector<int>& out) {
    out.clear();
    for (int s : set)
      for (auto& e : adj[s])
        if (!used[e.second] && in_set[e.first] && pairs[2 * e.second] == s)   // each pair once: from its first-listed station
          out.push_back(e.second);
  };
  auto emit = [&](const std::vector<int>& g) {
    for (int q : g) {
      used[q] = 1;
      const int a = pairs[2 * q], b = pairs[2 * q + 1];
      deg[a] -= 1;
      if (b != a) deg[b] -= 1;
    }
    groups.push_back(g);
  };
  std::vector<int> set, got;
  // 1. lines of AG(2, 8)
  if (S <= 64 && Bmax >= 4) {
    for (int line = 0; line < 72; ++line) {
      set.clear();
      for (int t = 0; t < 8; ++t) {
        int x, y;
        if (line < 64) { x = t; y = gf8_mul(line >> 3, t) ^ (line & 7); }
        else { x = line - 64; y = t; }
        const int s = 8 * x + y;
        if (s < S) set.push_back(s);
      }
      std::sort(set.begin(), set.end());
      for (int s : set) in_set[s] = 1;
      collect(set, got);
      for (int s : set) in_set[s] = 0;
      if ((int)got.size() <= Bmax && 4 * (int)got.size() >= 3 * Bmax) emit(got);
    }
  }
  // 2. greedy station sets
  std::vector<int> gain(S);
  long remaining = 0;
  for (int q = 0; q < P; ++q) remaining += !used[q];
  while (remaining > 0) {
    int seed = 0;
    for (int s = 1; s < S; ++s)
      if (deg[s] > deg[seed]) seed = s;
    std::fill(gain.begin(), gain.end(), 0);
    set.clear();
    set.push_back(seed);
    in_set[seed] = 1;
    int ne = 0;
    for (auto& e : adj[seed])
      if (!used[e.second]) { if (e.first == seed) ne += 1; else gain[e.first] += 1; }
    for (;;) {
      int best = -1;
      for (int s = 0; s < S; ++s)
        if (!in_set[s] && gain[s] > 0 && ne + gain[s] <= Bmax &&
            (best < 0 || gain[s] > gain[best] || (gain[s] == gain[best] && deg[s] > deg[best])))
          best = s;
      if (best < 0) break;
      if (set.size() >= 2 && 2 * gain[best] * (int)set.size() < ne) break;     // gain < half the current ratio
      set.push_back(best);
      in_set[best] = 1;
      ne += gain[best];
      for (auto& e : adj[best])
        if (!used[e.second]) { if (e.first == best) ne += 1; else if (!in_set[e.first]) gain[e.first] += 1; }
      if (ne >= Bmax) break;
    }
    std::sort(set.begin(), set.end());
    collect(set, got);
    for (int s : set) in_set[s] = 0;
    if ((int)got.size() > Bmax) got.resize(Bmax);     // self pairs can push a set over the limit
    if (got.empty())                                  // e.g. one pair listed more than Bmax times: take one
      for (auto& e : adj[seed])
        if (!used[e.second]) { got.push_back(e.second); break; }
    emit(got);
    remaining -= (long)got.size();
  }
  // 3. merge small groups (in order of appearance), then lay the order out
  std::vector<std::vector<int>> merged;
  std::vector<int> acc;
  for (auto& g : groups) {
    if (2 * (int)g.size() >= Bmax) { merged.push_back(g); continue; }
    if ((int)(acc.size() + g.size()) > Bmax) { merged.push_back(acc); acc.clear(); }
    acc.insert(acc.end(), g.begin(), g.end());
  }
  if (!acc.empty()) merged.push_back(acc);
  for (auto& g : merged) {
    bounds.push_back((int)perm.size());
    perm.insert(perm.end(), g.begin(), g.end());
  }
  bounds.push_back((int)perm.size());
}

}  // namespace xcx

// Host-side generator of the Hilbert transducer tables used by the device kernels (device_keys.hpp).
//
// The per-level update of the reference's iHilbert (R/sfc/hilbert.hpp:58-107) is replayed here on
// SYMBOLIC axes: a state records, for each of the three working coordinates, which original axis
// it currently holds and whether that axis is bit-flipped.  Breadth-first enumeration from the
// identity state yields every state the recurrence can reach together with the transitions.
#pragma once

#include <array>
#include <cstdint>
#include <map>
#include <tuple>
#include <vector>

#include "device_keys.hpp"

namespace cship
{

inline HilbertTables makeHilbertTables(int* numStatesOut = nullptr)
{
    struct State
    {
        std::array<int, 3> axis; // working coordinate a holds original axis axis[a] (0=x,1=y,2=z)
        std::array<int, 3> flip; // ... XOR all-ones if flip[a]
        bool operator<(const State& o) const { return std::tie(axis, flip) < std::tie(o.axis, o.flip); }
    };
    static constexpr unsigned octantToDigit[8] = {0, 1, 3, 2, 7, 6, 4, 5}; // R/sfc/hilbert.hpp:49

    std::map<State, int> ids;
    std::vector<State> states;
    auto idOf = [&](const State& s)
    {
        auto it = ids.find(s);
        if (it != ids.end()) return it->second;
        int id = int(states.size());
        if (id >= 48) return -1; // cannot happen: the group of signed axis permutations has 48 elements
        ids[s] = id;
        states.push_back(s);
        return id;
    };
    idOf(State{{0, 1, 2}, {0, 0, 0}});

    HilbertTables t{};
    for (size_t si = 0; si < states.size(); ++si)
    {
        State s = states[si];
        for (unsigned oct = 0; oct < 8; ++oct)
        {
            unsigned orig[3] = {(oct >> 2) & 1u, (oct >> 1) & 1u, oct & 1u};
            unsigned xi = orig[s.axis[0]] ^ s.flip[0];
            unsigned yi = orig[s.axis[1]] ^ s.flip[1];
            unsigned zi = orig[s.axis[2]] ^ s.flip[2];
            unsigned digit = octantToDigit[(xi << 2) | (yi << 1) | zi];

            // reflections applied to the remaining low bits, R/sfc/hilbert.hpp:85-87
            int fx = xi & ((!yi) | zi);
            int fy = (xi & (yi | zi)) | (yi & (!zi));
            int fz = (xi & (!yi) & (!zi)) | (yi & (!zi));
            State q = s;
            q.flip[0] ^= fx, q.flip[1] ^= fy, q.flip[2] ^= fz;
            // axis permutation, R/sfc/hilbert.hpp:89-103
            State n = q;
            if (zi)
            {
                n.axis = {q.axis[1], q.axis[2], q.axis[0]};
                n.flip = {q.flip[1], q.flip[2], q.flip[0]};
            }
            else if (!yi)
            {
                n.axis = {q.axis[2], q.axis[1], q.axis[0]};
                n.flip = {q.flip[2], q.flip[1], q.flip[0]};
            }
            int next = idOf(n);
            t.enc[si * 8 + oct]   = uint16_t(digit | (next << 3));
            t.dec[si * 8 + digit] = uint16_t(oct | (next << 3));
        }
    }
    // two levels per lookup: index = xx | yy << 2 | zz << 4 (the two bits of a coordinate: upper level, lower level)
    for (size_t si = 0; si < states.size() && si < size_t(HILBERT_STATES); ++si)
    {
        for (unsigned idx = 0; idx < 64; ++idx)
        {
            const unsigned xx = idx & 3u, yy = (idx >> 2) & 3u, zz = (idx >> 4) & 3u;
            const unsigned octHi = ((xx >> 1) << 2) | ((yy >> 1) << 1) | (zz >> 1);
            const unsigned octLo = ((xx & 1u) << 2) | ((yy & 1u) << 1) | (zz & 1u);
            const unsigned e1 = t.enc[si * 8 + octHi];
            const unsigned e2 = t.enc[(e1 >> 3) * 8 + octLo];
            t.enc2[si * 64 + idx] = uint16_t(((e1 & 7u) << 3) | (e2 & 7u) | ((e2 >> 3) << 6));
        }
    }
    if (numStatesOut) *numStatesOut = int(states.size());
    return t;
}

} // namespace cship

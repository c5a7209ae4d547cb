// bind.hpp — internal helper of the host layer: builds the binding list of one potential in the order of the reference's mws.make_* calls.
#pragma once
#include "sim.hpp"

namespace mistark {
struct BindList
{
    std::vector<mistark_binding> b;
    mistark_ctx* ctx;
    Stark& stark;
    BindList(mistark_ctx* c, Stark& s) : ctx(c), stark(s) {}
    void add(const double* host, int64_t n_items, int stride, int col)
    {
        const int id = mistark_array(ctx, host, n_items, stride);
        stark.check(id);
        b.push_back({id, stride, col});
    }
    void add_id(int id, int stride, int col) { b.push_back({id, stride, col}); }
    template <std::size_t N>
    void potential(const char* name, const std::vector<std::array<int32_t, N>>& conn)
    {
        stark.check(mistark_potential(ctx, name, conn.empty() ? nullptr : conn[0].data(), (int32_t)conn.size(), (int32_t)N, b.data(), (int32_t)b.size()));
    }
};
}  // namespace mistark

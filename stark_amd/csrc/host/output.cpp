// host/output.cpp — frame output of the host layer: the reference's DeformablesMeshOutput / RigidBodiesMeshOutput
// (stark/src/models/deformables/DeformablesMeshOutput.cpp:6-140, rigidbodies/RigidBodiesMeshOutput.cpp) write one VTK file per output label
// and frame through the write_frame callback (Stark.cpp:314-338); so does this, from the host mirror of the state. File names follow
// Stark::get_frame_path (<dir>/<simulation>_<label>_<frame>.vtk), points are stored as float like the reference's files
// (mesh_utils.cpp:122-183), so existing ParaView / pystark tooling reads them unchanged.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>

#include "sim.hpp"

namespace mistark {

namespace {
void put_be32(std::vector<char>& out, const void* v)
{
    const unsigned char* b = static_cast<const unsigned char*>(v);
    out.push_back((char)b[3]);
    out.push_back((char)b[2]);
    out.push_back((char)b[1]);
    out.push_back((char)b[0]);
}
}  // namespace

// legacy VTK, binary (big endian), unstructured grid: cell type 1 / 3 / 5 / 10 for 1 / 2 / 3 / 4 nodes per cell
void write_VTK(const std::string& path, const std::vector<Vec3>& vertices, const int* conn, size_t n_cells, int npc)
{
    static const int cell_type[5] = {0, 1, 3, 5, 10};
    if (npc < 1 || npc > 4) throw std::runtime_error("write_VTK: cells of 1 to 4 nodes");
    std::vector<char> out;
    char head[256];
    std::snprintf(head, sizeof(head), "# vtk DataFile Version 4.2\nmistark frame\nBINARY\nDATASET UNSTRUCTURED_GRID\nPOINTS %zu float\n", vertices.size());
    out.insert(out.end(), head, head + std::strlen(head));
    for (const Vec3& v : vertices)
        for (int k = 0; k < 3; k++) {
            const float f = (float)v[k];
            put_be32(out, &f);
        }
    std::snprintf(head, sizeof(head), "\nCELLS %zu %zu\n", n_cells, n_cells * (size_t)(npc + 1));
    out.insert(out.end(), head, head + std::strlen(head));
    for (size_t c = 0; c < n_cells; c++) {
        const int32_t n = npc;
        put_be32(out, &n);
        for (int k = 0; k < npc; k++) {
            const int32_t i = conn[c * npc + k];
            put_be32(out, &i);
        }
    }
    std::snprintf(head, sizeof(head), "\nCELL_TYPES %zu\n", n_cells);
    out.insert(out.end(), head, head + std::strlen(head));
    for (size_t c = 0; c < n_cells; c++) {
        const int32_t t = cell_type[npc];
        put_be32(out, &t);
    }
    out.push_back('\n');
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("write_VTK: cannot open '" + path + "'");
    const size_t w = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (w != out.size()) throw std::runtime_error("write_VTK: short write to '" + path + "'");
}

MeshOutput::MeshOutput(Stark& s, spPointDynamics d, spRigidBodyDynamics r) : stark(s), dyn(d), rb(r)
{
    stark.callbacks->add_write_frame([this]() { _write_frame(); });
}
void MeshOutput::add(const std::string& label, int npc, int point_set, int rigid_body, const std::vector<Vec3>& loc, const int* conn, size_t n)
{
    Mesh m;
    m.label = label;
    m.nodes_per_cell = npc;
    m.point_set = point_set;
    m.rigid_body = rigid_body;
    m.local_vertices = loc;
    m.conn.assign(conn, conn + n * npc);
    meshes.push_back(std::move(m));
}
void MeshOutput::add_point_set(const std::string& label, const PointSetHandler& set)
{
    std::vector<int> conn((size_t)set.size());
    for (size_t i = 0; i < conn.size(); i++) conn[i] = (int)i;
    add(label, 1, set.get_idx(), -1, {}, conn.data(), conn.size());
}
void MeshOutput::add_segment_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 2>>& conn)
{
    add(label, 2, set.get_idx(), -1, {}, conn.empty() ? nullptr : conn[0].data(), conn.size());
}
void MeshOutput::add_triangle_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 3>>& conn)
{
    add(label, 3, set.get_idx(), -1, {}, conn.empty() ? nullptr : conn[0].data(), conn.size());
}
void MeshOutput::add_triangle_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 3>>& conn, const std::vector<int>& point_set_map)
{
    add(label, 3, set.get_idx(), -1, {}, conn.empty() ? nullptr : conn[0].data(), conn.size());
    meshes.back().point_set_map = point_set_map;
}
void MeshOutput::add_tet_mesh(const std::string& label, const PointSetHandler& set, const std::vector<std::array<int, 4>>& conn)
{
    add(label, 4, set.get_idx(), -1, {}, conn.empty() ? nullptr : conn[0].data(), conn.size());
}
void MeshOutput::add_triangle_mesh(const std::string& label, const RigidBodyHandler& body, const std::vector<Vec3>& local_vertices, const std::vector<std::array<int, 3>>& conn)
{
    add(label, 3, -1, body.get_idx(), local_vertices, conn.empty() ? nullptr : conn[0].data(), conn.size());
}
void MeshOutput::_write_frame()
{
    if (meshes.empty()) return;
    if (stark.settings.output.output_directory.empty()) throw std::runtime_error("frame writes are enabled but Settings::output.output_directory is empty");
    dyn->mirror_to_host();  // one download of x0 per frame (the step loop itself never copies positions)
    // meshes sharing (label, cell size) go to one file (DeformablesMeshOutput.cpp:103-135)
    std::map<std::pair<std::string, int>, std::vector<const Mesh*>> groups;
    for (const Mesh& m : meshes) groups[{m.label, m.nodes_per_cell}].push_back(&m);
    for (const auto& g : groups) {
        std::vector<Vec3> V;
        std::vector<int> conn;
        for (const Mesh* m : g.second) {
            const int off = (int)V.size();
            if (m->point_set >= 0 && !m->point_set_map.empty()) {
                for (int loc : m->point_set_map) V.push_back(dyn->x1[dyn->get_begin(m->point_set) + loc]);
            } else if (m->point_set >= 0) {
                for (int i = dyn->get_begin(m->point_set); i < dyn->get_end(m->point_set); i++) V.push_back(dyn->x1[i]);
            } else {
                for (const Vec3& x : m->local_vertices) V.push_back(rb->get_position_at(m->rigid_body, x));
            }
            for (int i : m->conn) conn.push_back(i + off);
        }
        write_VTK(stark.get_frame_path(g.first.first) + ".vtk", V, conn.data(), conn.size() / g.first.second, g.first.second);
    }
    frames_written++;
}

}  // namespace mistark

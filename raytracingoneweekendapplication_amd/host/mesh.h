// mesh.h -- Wavefront OBJ ingestion behind the reference's `mesh` class (API of mesh.h:14-165: `loadObj(path, world,
// material, transform)`, `mesh_matrices`, `applyTransform/scale/rotate/translate`).  Host-side asset loading, outside
// the hot path; it exists so that scene code using `mesh::loadObj` compiles unchanged and yields
// the same triangle list as the reference's loader does for a well-formed file.
//
// Structure (not the reference's): the file is read into memory once and scanned by a cursor-based tokenizer
// (rtk_obj::scanner) into an indexed model -- positions, UVs and, per face, resolved corner indices -- with every
// index validated while it is parsed; emission into `world` is a separate pass over that model.
//
// Semantics a caller of the reference relies on, kept (behaviour of mesh.h:22-135, verified byte for byte on its
// monkey.obj by tests/test_abi_and_host.py):
//   * records: `v x y z`, `vt u v`, `f` with 3 or 4 corners written `v`, `v/vt`, `v//vn` or `v/vt/vn`; anything else
//     (normals, groups, materials, comments) is ignored, the .mtl file is never opened;
//   * a quad a b c d becomes the triangles (a, b, c) and (a, c, d); BOTH take the UVs of the face's first three
//     corners (the reference indexes the face's UV list with 0, 1, 2 in either call, mesh.h:79-80,129-131);
//   * faces with five or more corners are dropped with a diagnostic;
//   * each vertex goes through the single-precision 4x4 `transform` (as a homogeneous point) before it becomes a
//     double-precision `triangle`; `mesh_matrices` receives one matrix per triangle whose first three columns are the
//     transformed corners.
// Where the reference has undefined behaviour this loader is defined: a corner without a `vt` index (or with one out
// of range) gets UV (0, 0); negative indices count back from the most recent vertex as the OBJ format specifies; a
// face with a position index that is zero or out of range is dropped with a diagnostic instead of being read out of
// bounds; missing numbers on a `v` / `vt` line read as 0.
#ifndef RTK_MESH_H
#define RTK_MESH_H

#include <array>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "glm_min.h"
#include "rtk_scene_api.h"

namespace rtk_obj {

struct corner {
    int position;  // index into model::positions, always valid
    int uv;        // index into model::uvs, or -1 for "no texture coordinate"
};

struct model {
    std::vector<glm::vec3> positions;
    std::vector<glm::vec2> uvs;
    std::vector<std::array<corner, 3>> triangles;  // in file order; a quad contributes two
    int dropped_faces = 0;
};

// A cursor over one text buffer.  Lines end at '\n' (a preceding '\r' counts as blank).
class scanner {
public:
    scanner(const char* begin, const char* end) : p_(begin), end_(end) {}
    bool at_end() const { return p_ >= end_; }
    void skip_blanks() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    bool at_line_end() {
        skip_blanks();
        return p_ >= end_ || *p_ == '\n';
    }
    void next_line() {
        while (p_ < end_ && *p_ != '\n') ++p_;
        if (p_ < end_) ++p_;
    }
    // The blank-delimited word at the cursor (empty at a line end); the cursor moves past it.
    std::string word() {
        skip_blanks();
        const char* start = p_;
        while (p_ < end_ && *p_ != ' ' && *p_ != '\t' && *p_ != '\r' && *p_ != '\n') ++p_;
        return std::string(start, p_);
    }
    // A decimal real; 0 when the line has no further number.
    float real() {
        const std::string w = word();
        if (w.empty()) return 0.0f;
        char* stop = nullptr;
        const float value = std::strtof(w.c_str(), &stop);
        return stop == w.c_str() ? 0.0f : value;
    }

private:
    const char* p_;
    const char* end_;
};

// One face corner, "v", "v/vt", "v//vn" or "v/vt/vn": the raw 1-based (or negative, relative) indices; 0 = absent.
inline void split_corner(const std::string& text, long& v, long& vt) {
    v = vt = 0;
    const char* s = text.c_str();
    char* stop = nullptr;
    v = std::strtol(s, &stop, 10);
    if (stop == s) {
        v = 0;
        return;
    }
    if (*stop != '/') return;
    const char* t = stop + 1;
    if (*t == '/' || *t == '\0') return;  // "v//vn": no texture index
    vt = std::strtol(t, &stop, 10);
    if (stop == t) vt = 0;
}

// 1-based / negative-relative OBJ index -> 0-based index into a table of `count` entries, or -1 when it names nothing.
inline int resolve(long raw, size_t count) {
    const long n = long(count);
    if (raw > 0 && raw <= n) return int(raw - 1);
    if (raw < 0 && -raw <= n) return int(n + raw);
    return -1;
}

inline void parse(const std::string& text, model& out, std::ostream& diag) {
    scanner in(text.data(), text.data() + text.size());
    std::vector<corner> face;
    for (long line = 1; !in.at_end(); ++line, in.next_line()) {
        const std::string tag = in.word();
        if (tag == "v") {
            glm::vec3 p;
            p.x = in.real();
            p.y = in.real();
            p.z = in.real();
            out.positions.push_back(p);
        } else if (tag == "vt") {
            glm::vec2 t;
            t.x = in.real();
            t.y = in.real();
            out.uvs.push_back(t);
        } else if (tag == "f") {
            face.clear();
            bool usable = true;
            while (!in.at_line_end()) {
                long v = 0, vt = 0;
                split_corner(in.word(), v, vt);
                const int pos = resolve(v, out.positions.size());
                if (pos < 0) usable = false;
                face.push_back(corner{pos, resolve(vt, out.uvs.size())});
            }
            if (face.size() < 3) continue;  // degenerate record: nothing to emit (the reference ignores it silently too)
            if (!usable) {
                diag << "obj line " << line << ": face refers to a vertex that does not exist; face dropped" << std::endl;
                out.dropped_faces++;
            } else if (face.size() == 3) {
                out.triangles.push_back({face[0], face[1], face[2]});
            } else if (face.size() == 4) {
                // fan split; the second triangle keeps the UVs of corners 0, 1, 2 (reference behaviour, see the header)
                out.triangles.push_back({face[0], face[1], face[2]});
                out.triangles.push_back({corner{face[0].position, face[0].uv}, corner{face[2].position, face[1].uv}, corner{face[3].position, face[2].uv}});
            } else {
                diag << "obj line " << line << ": polygons with " << face.size() << " corners are not supported; face dropped" << std::endl;
                out.dropped_faces++;
            }
        }
    }
}

}  // namespace rtk_obj

class mesh {
public:
    mesh() {}
    mesh(const std::vector<glm::mat4>& triangles) : mesh_matrices(triangles) {}

    bool loadObj(const std::string path, hittable_list& world, const shared_ptr<lambertian> mat, glm::mat4 transform) {
        std::ifstream file(path, std::ios::binary);
        if (!file) {
            std::cerr << "mesh::loadObj: cannot read " << path << std::endl;
            return false;
        }
        const std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
        rtk_obj::model m;
        rtk_obj::parse(text, m, std::cerr);
        mesh_matrices.reserve(mesh_matrices.size() + m.triangles.size());
        for (const auto& tri : m.triangles) {
            glm::vec4 p[3];
            glm::vec2 uv[3];
            for (int k = 0; k < 3; k++) {
                p[k] = transform * glm::vec4(m.positions[size_t(tri[k].position)], 1.0f);
                uv[k] = tri[k].uv >= 0 ? m.uvs[size_t(tri[k].uv)] : glm::vec2(0, 0);
            }
            glm::mat4 columns(1.0f);
            columns[0] = p[0];
            columns[1] = p[1];
            columns[2] = p[2];
            columns[3] = glm::vec4(0, 0, 0, 1);
            mesh_matrices.push_back(columns);
            world.add(make_shared<triangle>(vec3(p[0].x, p[0].y, p[0].z), vec3(p[1].x, p[1].y, p[1].z), vec3(p[2].x, p[2].y, p[2].z), mat, uv[0], uv[1], uv[2]));
        }
        return true;
    }

    // The stored per-triangle matrices under a further transform (columns 0..2 are the corners).
    void applyTransform(const glm::mat4& transform) {
        for (glm::mat4& columns : mesh_matrices)
            for (int k = 0; k < 3; k++) columns[k] = transform * columns[k];
    }
    void scale(float factor) { applyTransform(glm::scale(glm::mat4(1.0f), glm::vec3(factor))); }
    void rotate(float angle, const glm::vec3& axis) { applyTransform(glm::rotate(glm::mat4(1.0f), glm::radians(angle), axis)); }
    void translate(const glm::vec3& offset) { applyTransform(glm::translate(glm::mat4(1.0f), offset)); }

    std::vector<glm::mat4> mesh_matrices;
};

#endif  // RTK_MESH_H

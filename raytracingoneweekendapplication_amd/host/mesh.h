// mesh.h -- OBJ ingestion with the reference's semantics (mesh.h:22-135): `v`, `vt` and `f v/vt/vn` records,
// triangles and quads (a quad becomes (0,1,2),(0,2,3)), every vertex pushed through a single-precision 4x4
// transform before it becomes a double-precision `triangle`.  Host-side asset loading, outside the hot path
// (SURVEY.md 8(f) row 2); it exists so that scene code using `mesh::loadObj` compiles and produces the same
// triangle list as the reference.
//
// Reference behaviours kept: both triangles of a quad take the face's FIRST three UV indices
// (mesh.h:79-80,129-131); faces with more than four vertices are skipped with a message; the `.mtl` file is
// never read.  Difference: a face record without a `vt` index gets UV (0,0) instead of reading an
// uninitialised index.
#ifndef RTK_MESH_H
#define RTK_MESH_H

#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "glm_min.h"
#include "rtk_scene_api.h"

class mesh {
public:
    mesh() {}
    mesh(const std::vector<glm::mat4>& triangles) : mesh_matrices(triangles) {}

    bool loadObj(const std::string path, hittable_list& world, const shared_ptr<lambertian> mat, glm::mat4 transform) {
        std::ifstream file(path);
        if (!file.is_open()) {
            std::cerr << "Failed to open file: " << path << std::endl;
            return false;
        }
        std::vector<glm::vec3> positions;
        std::vector<glm::vec2> uvs;
        std::string line;
        while (std::getline(file, line)) {
            std::istringstream ss(line);
            std::string tag;
            ss >> tag;
            if (tag == "v") {
                glm::vec3 p;
                ss >> p.x >> p.y >> p.z;
                positions.push_back(p);
            } else if (tag == "vt") {
                glm::vec2 t;
                ss >> t.x >> t.y;
                uvs.push_back(t);
            } else if (tag == "f") {
                std::vector<int> vi, ti;
                std::string group;
                while (ss >> group) {
                    std::istringstream gs(group);
                    int v = 0, vt = 0, vn = 0;
                    char slash;
                    gs >> v >> slash >> vt >> slash >> vn;
                    vi.push_back(v - 1);
                    ti.push_back(vt - 1);
                }
                if (vi.size() < 3) continue;
                if (vi.size() == 3) {
                    add_triangle(positions, uvs, vi[0], vi[1], vi[2], ti, mat, world, transform);
                } else if (vi.size() == 4) {
                    add_triangle(positions, uvs, vi[0], vi[1], vi[2], ti, mat, world, transform);
                    add_triangle(positions, uvs, vi[0], vi[2], vi[3], ti, mat, world, transform);
                } else {
                    std::cerr << "Skipping face with " << vi.size() << " vertices." << std::endl;
                }
            }
        }
        return true;
    }

    void applyTransform(const glm::mat4& transform) {
        for (auto& m : mesh_matrices)
            for (int i = 0; i < 3; ++i) m[i] = transform * m[i];
    }
    void scale(float factor) { applyTransform(glm::scale(glm::mat4(1.0f), glm::vec3(factor))); }
    void rotate(float angle, const glm::vec3& axis) { applyTransform(glm::rotate(glm::mat4(1.0f), glm::radians(angle), axis)); }
    void translate(const glm::vec3& offset) { applyTransform(glm::translate(glm::mat4(1.0f), offset)); }

    std::vector<glm::mat4> mesh_matrices;  // one matrix per triangle: columns = transformed vertices (mesh.h:112-118)

private:
    void add_triangle(const std::vector<glm::vec3>& positions, const std::vector<glm::vec2>& uvs, int a, int b, int c, const std::vector<int>& ti,
                      const shared_ptr<lambertian> mat, hittable_list& world, const glm::mat4& transform) {
        const glm::vec4 p0 = transform * glm::vec4(positions[a], 1.0f);
        const glm::vec4 p1 = transform * glm::vec4(positions[b], 1.0f);
        const glm::vec4 p2 = transform * glm::vec4(positions[c], 1.0f);
        glm::mat4 m(1.0f);
        m[0] = p0; m[1] = p1; m[2] = p2; m[3] = glm::vec4(0, 0, 0, 1);
        mesh_matrices.push_back(m);
        auto uv_at = [&](int k) { return (ti[k] >= 0 && ti[k] < int(uvs.size())) ? uvs[ti[k]] : glm::vec2(0, 0); };
        world.add(make_shared<triangle>(vec3(p0.x, p0.y, p0.z), vec3(p1.x, p1.y, p1.z), vec3(p2.x, p2.y, p2.z), mat, uv_at(0), uv_at(1), uv_at(2)));
    }
};

#endif  // RTK_MESH_H

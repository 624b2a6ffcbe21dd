// rt_obj.cpp — OBJ ingest, mirroring src/obj_file.zig: `parseObjFile` (:77-147),
// `triangulatePolygon` (:156-164) and `ObjFile.toShapeGroup` (:45-62).  Host only; runs once
// per asset and feeds the scene model the flattener consumes.
//
// Parser rules kept from the reference: lines split on '\n', literals on ' '; a command is a
// single character `v`, `f` or `g` (anything longer or different is InvalidCommand, :101,137);
// `v` takes three floats; `f` takes every remaining literal as a 1-based vertex index (so
// "1/2/3" is an error) and fan-triangulates more than three of them — including the
// degenerate first triangle (v0, v0, v1) the reference's loop produces (:158-163); `g` takes
// one name; a left-over literal is InvalidCommand (:141).  Faces before any `g` go to the
// group's own triangle list, the others to one identity subgroup per name.  The reference
// iterates a StringHashMap there (:54), whose order is unspecified; here subgroups appear in
// order of first use (this only matters for exact t ties between different named groups).
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rt.hpp"

using namespace rt;

struct rt_obj {
    std::vector<tup::Point> vertices;  // index 0 is the unused sentinel (:86-89)
    std::vector<Triangle> faces;       // default group
    std::vector<std::string> group_names;
    std::vector<std::vector<Triangle>> group_faces;
    // toShapeGroup result
    Group shape_group;
    GroupView view;
    const rt_group* c_group = nullptr;
};

namespace {

std::vector<std::string> tokenize(const std::string& s, char sep) {  // std.mem.tokenize: empty tokens skipped
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && s[i] == sep) ++i;
        size_t j = i;
        while (j < s.size() && s[j] != sep) ++j;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

bool parse_float(const std::string& s, float& out) {
    if (s.empty()) return false;
    char* end = nullptr;
    errno = 0;
    out = std::strtof(s.c_str(), &end);
    return end == s.c_str() + s.size();
}

bool parse_index(const std::string& s, size_t& out) {  // std.fmt.parseInt(usize, s, 10)
    if (s.empty()) return false;
    size_t i = s[0] == '+' ? 1 : 0;
    if (i == s.size()) return false;
    size_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (size_t)(s[i] - '0');
    }
    out = v;
    return true;
}

Group shape_group_of(const rt_obj& o) {  // toShapeGroup, :45-62
    Group g = Group::init(mat::identity());
    g.triangles = o.faces;
    for (const auto& faces : o.group_faces) {
        if (faces.empty()) continue;
        Group sub = Group::init(mat::identity());
        sub.triangles = faces;
        g.subgroups.push_back(std::move(sub));
    }
    return g;
}

}  // namespace

namespace rt {
namespace obj {

ObjFile parseObjFile(const std::string& text) {
    rt_obj* h = nullptr;
    const int rc = rt_obj_parse(text.data(), text.size(), &h);
    if (rc != 0)
        throw ParserError(rc, rc == RT_OBJ_INVALID_COMMAND ? "InvalidCommand" : rc == RT_OBJ_BAD_NUMBER ? "bad number" : "bad vertex index");
    return ObjFile{std::shared_ptr<rt_obj>(h, rt_obj_free)};
}

Group ObjFile::toShapeGroup() const { return shape_group_of(*handle); }

}  // namespace obj
}  // namespace rt

extern "C" {

int rt_obj_parse(const char* text, uint64_t len, rt_obj** out) {
    if (!text || !out) return RTX_E_INVALID;
    *out = nullptr;
    rt_obj* obj = new rt_obj();
    obj->vertices.push_back(tup::point(0, 0, 0));
    int err = 0;
    int group = -1;  // index into group_names; -1 = default list
    for (const std::string& line : tokenize(std::string(text, (size_t)len), '\n')) {
        const std::vector<std::string> lit = tokenize(line, ' ');
        if (lit.empty()) continue;
        if (lit[0].size() > 1) { err = RT_OBJ_INVALID_COMMAND; break; }
        size_t used = 1;
        if (lit[0][0] == 'v') {
            float xyz[3];
            for (int k = 0; k < 3 && !err; ++k) {
                if (used >= lit.size()) err = RT_OBJ_INVALID_COMMAND;  // getLiteral, :149-151
                else if (!parse_float(lit[used++], xyz[k])) err = RT_OBJ_BAD_NUMBER;
            }
            if (err) break;
            obj->vertices.push_back(tup::point(xyz[0], xyz[1], xyz[2]));
        } else if (lit[0][0] == 'f') {
            std::vector<size_t> idx;
            for (; used < lit.size() && !err; ++used) {
                size_t v;
                if (!parse_index(lit[used], v)) err = RT_OBJ_BAD_NUMBER;
                else idx.push_back(v);
            }
            if (err) break;
            if (idx.size() < 3) { err = RT_OBJ_BAD_INDEX; break; }
            for (size_t v : idx)
                if (v >= obj->vertices.size()) err = RT_OBJ_BAD_INDEX;
            if (err) break;
            std::vector<Triangle>& faces = group < 0 ? obj->faces : obj->group_faces[(size_t)group];
            const std::vector<tup::Point>& V = obj->vertices;
            if (idx.size() > 3) {
                // triangulatePolygon (:156-164): for (indices[1..]) |_, i| (v[idx0], v[idx[i]], v[idx[i+1]]), i from 0
                for (size_t i = 0; i + 1 < idx.size(); ++i) faces.push_back(Triangle::init(V[idx[0]], V[idx[i]], V[idx[i + 1]]));
            } else {
                faces.push_back(Triangle::init(V[idx[0]], V[idx[1]], V[idx[2]]));
            }
        } else if (lit[0][0] == 'g') {
            if (used >= lit.size()) { err = RT_OBJ_INVALID_COMMAND; break; }
            const std::string& name = lit[used++];
            group = -1;
            for (size_t k = 0; k < obj->group_names.size(); ++k)
                if (obj->group_names[k] == name) group = (int)k;
            if (group < 0) {  // the reference creates the entry at the first face; an empty group adds nothing either way
                obj->group_names.push_back(name);
                obj->group_faces.emplace_back();
                group = (int)obj->group_names.size() - 1;
            }
        } else {
            err = RT_OBJ_INVALID_COMMAND;
            break;
        }
        if (used < lit.size()) { err = RT_OBJ_INVALID_COMMAND; break; }  // unused literal, :141
    }
    if (err) {
        delete obj;
        return err;
    }
    *out = obj;
    return 0;
}

uint32_t rt_obj_vertex_count(const rt_obj* o) { return (uint32_t)o->vertices.size() - 1; }
uint32_t rt_obj_face_count(const rt_obj* o) {
    size_t n = o->faces.size();
    for (const auto& g : o->group_faces) n += g.size();
    return (uint32_t)n;
}
uint32_t rt_obj_group_count(const rt_obj* o) {
    uint32_t n = 0;
    for (const auto& g : o->group_faces) n += g.empty() ? 0u : 1u;
    return n;
}

// toShapeGroup (:45-62) followed by Group.bound() when `bound` is set (samples/teapot.zig:19).
int rt_obj_to_group(rt_obj* o, int bound, const rt_group** out) {
    if (!o || !out) return RTX_E_INVALID;
    Group g = shape_group_of(*o);
    if (bound) g.bound();
    o->shape_group = std::move(g);
    o->view = GroupView();
    std::vector<Group> one;
    one.push_back(o->shape_group);  // GroupView points into the vectors it is given: keep this copy alive
    o->view.keep = std::move(one);
    o->c_group = o->view.build(o->view.keep);
    *out = o->c_group;
    return 0;
}

void rt_obj_free(rt_obj* o) { delete o; }

}  // extern "C"

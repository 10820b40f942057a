// scene.cpp -- bodies, camera and the objects.txt loader (host side).
//
// Behaviour follows /root/reference/source/scene.cpp:92-190 (loader),
// body.cpp:114-169 (generators) and object.cpp:18-42 (camera).  The generated
// primitive centres and the camera matrix are inputs of the device kernels, so
// the fp32 operation order of the reference is kept (this file is compiled with
// -ffp-contract=off).
#include "scene.h"

#include <cmath>
#include <fstream>
#include <sstream>

namespace Object {

Camera::Camera(float3 position, float3 direction, float3 up, float FOV)
    : position(position), direction(direction), up(up), FOV(FOV), focal(0.0f) {
    update();
}

// object.cpp:23-36: orthonormal basis in the columns (right, up', -forward, position).
void Camera::update() {
    const float3 right = rmath::normalize(rmath::cross(direction, up));
    const float3 upward = rmath::normalize(rmath::cross(right, direction));
    const float3 forward = rmath::normalize(direction);

    transform.set_col(0, right.x, right.y, right.z, 0.0f);
    transform.set_col(1, upward.x, upward.y, upward.z, 0.0f);
    transform.set_col(2, -forward.x, -forward.y, -forward.z, 0.0f);
    transform.set_col(3, position.x, position.y, position.z, 1.0f);

    // The reference calls the unqualified tan() on a float, which takes the
    // double overload; the product with 2.0f is then rounded to fp32.
    const float half_angle = FOV * rmath::DEG_TO_RAD / 2;
    focal = 2.0f * static_cast<float>(std::tan(static_cast<double>(half_angle)));
}

// object.cpp:38-42
float3 Camera::view(float3 vector, bool offset) const {
    return rmath::transform(transform, vector, offset ? 1.0f : 0.0f);
}

}  // namespace Object

namespace Body {

namespace {

// Sub-cube order of one Menger subdivision step, in units of the sub-cube edge:
// the z = -1 ring, the z = +1 ring, then the four z = 0 corners -- the order in
// which body.cpp:122-145 recurses.  Only cubes with at most one zero coordinate.
struct Offset { int x, y, z; };
const Offset kRing[8] = { { 1, -1, 0 }, { 0, -1, 0 }, { -1, -1, 0 }, { 1, 1, 0 },
                          { 0, 1, 0 },  { -1, 1, 0 }, { -1, 0, 0 },  { 1, 0, 0 } };
const Offset kMiddle[4] = { { -1, -1, 0 }, { 1, -1, 0 }, { -1, 1, 0 }, { 1, 1, 0 } };

inline float signed_step(int s, float d) { return s > 0 ? d : (s < 0 ? -d : 0.0f); }

void subdivide(List *out, float3 center, float edge, int iterations, float3 color) {
    const float d = edge / 3;
    out->append(new Cross(center, float3(d), color));
    if (iterations < 2) return;
    for (int ring = 0; ring < 2; ring++) {
        const int z = ring == 0 ? -1 : 1;
        for (const Offset &o : kRing) {
            const float3 step(signed_step(o.x, d), signed_step(o.y, d), signed_step(z, d));
            subdivide(out, center + step, d, iterations - 1, color);
        }
    }
    for (const Offset &o : kMiddle) {
        const float3 step(signed_step(o.x, d), signed_step(o.y, d), 0.0f);
        subdivide(out, center + step, d, iterations - 1, color);
    }
}

}  // namespace

// body.cpp:149-156: the solid cube minus every generated cross.
List *MengerSponge(float3 position, float size, int iterations, float3 color) {
    List *sponge = new List(Mode::DIFFERENCE);
    sponge->append(new Box(position, float3(size), color));
    subdivide(sponge, position, size, iterations, color);
    return sponge;
}

// body.cpp:159-169: a sphere minus an equal sphere shifted by 1.5 radii along +x.
List *DeathStar(float3 position, float radius, float3 color) {
    List *star = new List(Mode::DIFFERENCE);
    star->append(new Sphere(position, radius, color));
    const float3 bite(position.x + 1.5f * radius, position.y, position.z);
    star->append(new Sphere(bite, radius, color));
    return star;
}

}  // namespace Body

namespace scene {

Scene::Scene() : tree(new Body::List()), camera(new Object::Camera()) {}

namespace {

float3 read3(std::istringstream &in) {
    float3 v;
    in >> v.x >> v.y >> v.z;
    return v;
}

}  // namespace

// scene.cpp:100-189.  Unknown commands and blank lines are skipped; "Color" is
// sticky and starts white; "Bounds s" is the complement of a black box of edge s.
void Scene::parse(const std::string &text) {
    std::istringstream stream(text);
    std::string line;
    float3 color(1.0f);
    while (std::getline(stream, line)) {
        std::istringstream in(line);
        std::string cmd;
        in >> cmd;
        Body::Base *body = nullptr;
        if (cmd == "Sphere") {
            const float3 position = read3(in);
            float radius = 0.0f;
            in >> radius;
            body = new Body::Sphere(position, radius, color);
        } else if (cmd == "Box" || cmd == "Cross") {
            const float3 position = read3(in);
            const float3 size = read3(in);
            if (cmd == "Box") body = new Body::Box(position, size, color);
            else body = new Body::Cross(position, size, color);
        } else if (cmd == "Bounds") {
            float size = 0.0f;
            in >> size;
            Body::List *shell = new Body::List(Body::Mode::COMPLEMENT);
            shell->append(new Body::Box(float3(0.0f), float3(size), float3(0.0f)));
            body = shell;
        } else if (cmd == "DeathStar") {
            const float3 position = read3(in);
            float radius = 0.0f;
            in >> radius;
            body = Body::DeathStar(position, radius, color);
        } else if (cmd == "MengerSponge") {
            const float3 position = read3(in);
            float size = 0.0f;
            int iterations = 0;
            in >> size >> iterations;
            body = Body::MengerSponge(position, size, iterations, color);
        }
        if (body) {
            tree->append(body);
            continue;
        }
        if (cmd == "Light") {
            lights.emplace_back(new Object::Light(read3(in)));
        } else if (cmd == "Camera") {
            std::string field;
            in >> field;
            if (field == "Position") camera->position = read3(in);
            else if (field == "Direction") camera->direction = read3(in);
            else if (field == "Up") camera->up = read3(in);
            else if (field == "FOV") in >> camera->FOV;
        } else if (cmd == "Color") {
            color = read3(in);
        }
    }
    camera->update();
}

bool Scene::load(const char *path) {
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) {
        parse(std::string());
        return false;
    }
    std::ostringstream text;
    text << file.rdbuf();
    parse(text.str());
    return true;
}

namespace {

void emit(const Body::Base *body, std::vector<rm_node> &out) {
    rm_node n = {};
    n.type = static_cast<uint32_t>(body->type);
    switch (body->type) {
    case Body::Type::LIST: {
        const Body::List *list = static_cast<const Body::List *>(body);
        n.mode = static_cast<uint32_t>(list->mode);
        n.child_count = static_cast<uint32_t>(list->bodies.size());
        out.push_back(n);
        for (const auto &kid : list->bodies) emit(kid.get(), out);
        return;
    }
    case Body::Type::SPHERE: {
        const Body::Sphere *s = static_cast<const Body::Sphere *>(body);
        n.center[0] = s->position.x; n.center[1] = s->position.y; n.center[2] = s->position.z;
        n.size[0] = s->radius;
        n.color[0] = s->color.x; n.color[1] = s->color.y; n.color[2] = s->color.z;
        break;
    }
    case Body::Type::BOX: {
        const Body::Box *b = static_cast<const Body::Box *>(body);
        n.center[0] = b->position.x; n.center[1] = b->position.y; n.center[2] = b->position.z;
        n.size[0] = b->size.x; n.size[1] = b->size.y; n.size[2] = b->size.z;
        n.color[0] = b->color.x; n.color[1] = b->color.y; n.color[2] = b->color.z;
        break;
    }
    case Body::Type::CROSS: {
        const Body::Cross *c = static_cast<const Body::Cross *>(body);
        n.center[0] = c->position.x; n.center[1] = c->position.y; n.center[2] = c->position.z;
        n.size[0] = c->size.x; n.size[1] = c->size.y; n.size[2] = c->size.z;
        n.color[0] = c->color.x; n.color[1] = c->color.y; n.color[2] = c->color.z;
        break;
    }
    }
    out.push_back(n);
}

}  // namespace

std::vector<rm_node> Scene::nodes() const {
    std::vector<rm_node> out;
    emit(tree.get(), out);
    return out;
}

std::vector<float> Scene::light_positions() const {
    std::vector<float> out;
    out.reserve(lights.size() * 3);
    for (const auto &l : lights) {
        out.push_back(l->position.x);
        out.push_back(l->position.y);
        out.push_back(l->position.z);
    }
    return out;
}

// ---- process-global scene, as the reference keeps it (scene.cpp:19-23) ----

Body::List *tree = nullptr;
std::vector<Object::Light *> lights;
Object::Camera *camera = nullptr;

namespace {
std::unique_ptr<Scene> g_scene;
}

Scene &current() {
    if (!g_scene) {
        g_scene.reset(new Scene());
        tree = g_scene->tree.get();
        camera = g_scene->camera.get();
        lights.clear();
    }
    return *g_scene;
}

void load(const char *path) {
    g_scene.reset(new Scene());
    g_scene->load(path);
    tree = g_scene->tree.get();
    camera = g_scene->camera.get();
    lights.clear();
    for (const auto &l : g_scene->lights) lights.push_back(l.get());
}

}  // namespace scene

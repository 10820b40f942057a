// scene.h -- host scene model: bodies, lights, camera and the objects.txt loader.
//
// Keeps the reference's host API for the part of the program that feeds the
// render path (names and argument meaning as in
// /root/reference/source/include/body.h, object.h, scene.h):
//   Body::{Type, Mode, Base, List, Sphere, Box, Cross, MengerSponge, DeathStar}
//   Object::{Light, Camera}
//   scene::{tree, lights, camera, load}
// The signed-distance evaluation itself lives on the device only; there is no
// host SDF in this library.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "vecmath.h"
#include "../../include/raymarch_hip.h"

namespace Object {

using rmath::float3;
using rmath::float4x4;

// object.h:21-25 -- the colour is carried but, as in the reference, unused by shading.
struct Light {
    float3 position;
    float3 color;
    explicit Light(float3 position, float3 color = float3(1.0f)) : position(position), color(color) {}
};

// object.h:27-41, object.cpp:18-42
struct Camera {
    float3 position;
    float3 direction;
    float3 up;
    float FOV;

    float4x4 transform;
    float focal;

    Camera(float3 position = float3(0.0f), float3 direction = float3(0.0f, 0.0f, -1.0f),
           float3 up = float3(0.0f, 1.0f, 0.0f), float FOV = 90);
    void update();
    float3 view(float3 vector, bool offset = true) const;
};

}  // namespace Object

namespace Body {

using rmath::float3;

enum class Type : uint32_t { LIST = 0, SPHERE = 1, BOX = 2, CROSS = 3 };
enum class Mode : uint32_t { UNION = 0, COMPLEMENT = 1, INTERSECTION = 2, DIFFERENCE = 3 };

struct Base {
    Type type;
    explicit Base(Type type) : type(type) {}
    virtual ~Base() = default;
};

// A list owns the bodies appended to it.
struct List : Base {
    std::vector<std::unique_ptr<Base>> bodies;
    Mode mode;
    explicit List(Mode mode = Mode::UNION) : Base(Type::LIST), mode(mode) {}
    void append(Base *body) { bodies.emplace_back(body); }
};

struct Sphere : Base {
    float3 position;
    float radius;
    float3 color;
    Sphere(float3 position, float radius, float3 color = float3(1.0f))
        : Base(Type::SPHERE), position(position), radius(radius), color(color) {}
};

struct Box : Base {
    float3 position;
    float3 size;
    float3 color;
    Box(float3 position, float3 size, float3 color = float3(1.0f))
        : Base(Type::BOX), position(position), size(size), color(color) {}
};

struct Cross : Base {
    float3 position;
    float3 size;
    float3 color;
    Cross(float3 position, float3 size, float3 color = float3(1.0f))
        : Base(Type::CROSS), position(position), size(size), color(color) {}
};

// Generators (body.h:74-83).  The caller owns the returned list.
List *MengerSponge(float3 position, float size, int iterations = 3, float3 color = float3(1.0f));
List *DeathStar(float3 position, float radius, float3 color = float3(1.0f));

}  // namespace Body

namespace scene {

using rmath::float3;

// A self-contained scene; the reference's process globals below wrap one.
struct Scene {
    std::unique_ptr<Body::List> tree;
    std::vector<std::unique_ptr<Object::Light>> lights;
    std::unique_ptr<Object::Camera> camera;

    Scene();
    void parse(const std::string &text);   // the body of scene::load
    bool load(const char *path);           // false if the file could not be opened (scene stays empty)

    // Depth-first pre-order listing of the tree for rm_set_scene.
    std::vector<rm_node> nodes() const;
    std::vector<float> light_positions() const;
};

// The reference's globals (scene.h:12-14).  load() replaces them.
extern Body::List *tree;
extern std::vector<Object::Light *> lights;
extern Object::Camera *camera;
void load(const char *path);
Scene &current();

}  // namespace scene

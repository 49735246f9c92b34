// Entity physics over batched AABB ray fans: the consumer of the picker path (src/systems/physics.rs).
// One step = one PickerBatch holding every entity's AABB -> one vx_raycast -> per-entity velocity clamping. The
// reference runs this at 250 Hz for a single entity with a blocking GL round trip per call (gamelogic/game.rs:90,
// svo.rs:248-249); step_many is the shape that amortises the round trip.
#pragma once

#include <cmath>
#include <vector>

#include "svo_picker.hpp"

namespace vx {
namespace systems {

constexpr float kPhysicsEpsilon = 0.0005f;  // physics.rs:8

// physics.rs:30-34
struct EntityState {
    bool is_grounded = false;  // colliding in -y direction
    bool operator==(const EntityState& o) const { return is_grounded == o.is_grounded; }
};

// physics.rs:36-57
struct EntityCapabilities {
    bool wall_clip = false;           // disables all collisions along x & z
    bool flying = false;              // disables gravity and all collisions
    float gravity = 60.0f;            // constant acceleration in -y
    float max_fall_velocity = 100.0f; // cap on the fall speed gravity can build up
};

// physics.rs:77-87
struct AABBDef {
    Vec3 offset, extents;
};

// physics.rs:10-28, 59-75
struct Entity {
    Vec3 position, velocity, euler_rotation;
    AABBDef aabb_def;
    EntityCapabilities caps;
    EntityState state;

    Entity() = default;
    Entity(Vec3 position_, AABBDef aabb_def_) : position(position_), aabb_def(aabb_def_) {}

    Vec3 get_forward() const {
        const Vec3 f{std::cos(euler_rotation.y) * std::cos(euler_rotation.x), std::sin(euler_rotation.x),
                     std::sin(euler_rotation.y) * std::cos(euler_rotation.x)};
        const float len = std::sqrt(f.x * f.x + f.y * f.y + f.z * f.z);
        return Vec3{f.x / len, f.y / len, f.z / len};
    }
    const EntityState& get_state() const { return state; }
};

// physics.rs:88-96: anything that answers a PickerBatch (graphics::Svo, systems::worldsvo::Svo, a test double)
struct Raycaster {
    virtual ~Raycaster() = default;
    virtual void raycast(PickerBatch& batch, PickerBatchResult& result) const = 0;
};

// adapts any object with raycast(PickerBatch&, PickerBatchResult&) const
template <class T>
struct RaycasterRef final : Raycaster {
    const T& target;
    explicit RaycasterRef(const T& t) : target(t) {}
    void raycast(PickerBatch& batch, PickerBatchResult& result) const override { target.raycast(batch, result); }
};

class Physics {
public:
    // physics.rs:111-118
    void step(float delta_time, const Raycaster& raycaster, Entity& entity) {
        reset();
        add_entity(entity);
        raycaster.raycast(batch_, result_);
        update_entity(entity, result_.aabbs.at(0), delta_time);
    }

    // physics.rs:122-136
    void step_many(float delta_time, const Raycaster& raycaster, std::vector<Entity>& entities) {
        reset();
        for (const Entity& e : entities) add_entity(e);
        raycaster.raycast(batch_, result_);
        for (size_t i = 0; i < entities.size(); ++i) update_entity(entities[i], result_.aabbs.at(i), delta_time);
    }

    // physics.rs:139-170
    static void update_entity(Entity& entity, const AabbResult& result, float delta_time) {
        // apply gravity
        if (!entity.caps.flying) {
            entity.velocity.y -= entity.caps.gravity * delta_time;
            if (entity.velocity.y < 0.0f) entity.velocity.y = std::fmax(entity.velocity.y, -entity.caps.max_fall_velocity);
        }
        Vec3 velocity{entity.velocity.x * delta_time, entity.velocity.y * delta_time, entity.velocity.z * delta_time};

        // entity state with the new velocity
        entity.state.is_grounded = !entity.caps.flying && (result.neg.y + velocity.y) < 0.02f && result.neg.y != -1.0f;
        // reset gravity if the entity stands on the ground already
        if (entity.state.is_grounded && entity.velocity.y < 0.0f) entity.velocity.y = 0.0f;

        // constrain the velocity by nearby collisions
        if (!entity.caps.flying) {
            if (!entity.caps.wall_clip) {
                velocity.x = apply_axial_physics(velocity.x, result.pos.x, result.neg.x);
                velocity.z = apply_axial_physics(velocity.z, result.pos.z, result.neg.z);
            }
            velocity.y = apply_axial_physics(velocity.y, result.pos.y, result.neg.y);
        }
        entity.position.x += velocity.x;
        entity.position.y += velocity.y;
        entity.position.z += velocity.z;
    }

    // physics.rs:173-185
    static float apply_axial_physics(float speed, float dst_pos, float dst_neg) {
        const float dst = speed > 0.0f ? dst_pos : dst_neg;
        if (dst == -1.0f) return speed;
        if (dst < 2.0f * kPhysicsEpsilon) return 0.0f;
        if (std::fabs(speed) > dst) return (dst - kPhysicsEpsilon) * signum(speed);
        return speed;
    }

    const PickerBatch& last_batch() const { return batch_; }

private:
    // f32::signum: 1.0 for +0.0 and positives, -1.0 for -0.0 and negatives, NaN for NaN
    static float signum(float v) { return std::isnan(v) ? v : (std::signbit(v) ? -1.0f : 1.0f); }
    void reset() { batch_.reset(); result_.reset(); }
    // physics.rs:205-208
    void add_entity(const Entity& e) { batch_.add_aabb(Aabb{e.position, e.aabb_def.offset, e.aabb_def.extents}); }

    PickerBatch batch_;   // reused between steps (physics.rs:98-100)
    PickerBatchResult result_;
};

}  // namespace systems
}  // namespace vx

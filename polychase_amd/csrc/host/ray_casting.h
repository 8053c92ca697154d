// ray_casting.h -- AcceleratedMesh and RayCast of the reference (cpp/ray_casting.h:23-51,
// cpp/ray_casting.cc) with the Embree scene replaced by a mesh resident on the GPU, an LBVH built on the
// GPU at construction, and a batched closest-hit kernel (pc_raycast_pixels).
#pragma once

#include <memory>
#include <optional>
#include <vector>

#include "types.h"

struct pc_mesh;

class AcceleratedMesh {
   public:
    AcceleratedMesh(std::vector<float> vertices, std::vector<uint32_t> triangles, std::vector<uint32_t> masked_triangles);
    AcceleratedMesh(const AcceleratedMesh&) = delete;
    AcceleratedMesh& operator=(const AcceleratedMesh&) = delete;
    ~AcceleratedMesh();

    const Mesh& Inner() const { return mesh_; }
    Mesh& InnerMut() { return mesh_; }
    pc_mesh* Gpu() const { return gpu_; }
    // sends the current masked_triangles bits to the GPU copy (they can be edited through InnerMut())
    void SyncMask() const;

    // Batched RayCast(accel_mesh, scene_transform, pos, check_mask): hits[i] is empty on a miss.
    // exhaustive: sweep over every triangle instead of walking the hierarchy (validation only)
    void RayCastPixels(const SceneTransformations& scene_transform, const float* xy, size_t n, bool check_mask,
                       std::vector<std::optional<RayHit>>& hits, bool exhaustive = false) const;

   private:
    Mesh mesh_;
    pc_mesh* gpu_ = nullptr;
    mutable std::vector<uint8_t> hit_;
    mutable std::vector<float> pos_, uvt_;
    mutable std::vector<uint32_t> prim_;
};

struct pc_ray_camera;
// GetRayObjectSpace (cpp/ray_casting.h:53-63) as the camera block the GPU kernels take
void MakeRayCamera(const SceneTransformations& scene_transform, pc_ray_camera* out);

std::optional<RayHit> RayCast(const AcceleratedMesh& accel_mesh, const SceneTransformations& scene_transform, Vec2f pos,
                              bool check_mask);

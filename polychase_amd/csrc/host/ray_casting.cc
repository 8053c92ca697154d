#include "ray_casting.h"

#include <stdexcept>

#include "gpu_context.h"

void WarmTrackerCaches();   // track_sequence.cc

AcceleratedMesh::AcceleratedMesh(std::vector<float> vertices, std::vector<uint32_t> triangles,
                                 std::vector<uint32_t> masked_triangles)
    : mesh_(std::move(vertices), std::move(triangles), std::move(masked_triangles)) {
    GpuSection section;
    pc_context* ctx = SharedGpuContext();
    if (pc_mesh_create(ctx, mesh_.vertices.data(), static_cast<int>(mesh_.NumVertices()), mesh_.triangles.data(),
                       static_cast<int>(mesh_.NumTriangles()), &gpu_) != PC_OK)
        throw std::runtime_error(std::string("pc_mesh_create: ") + pc_last_error());
    // tracking always follows the construction of a mesh: what its first call would allocate is allocated now (track_sequence.h)
    WarmTrackerCaches();
}

AcceleratedMesh::~AcceleratedMesh() {
    GpuSection section;
    pc_mesh_destroy(gpu_);
}

void AcceleratedMesh::SyncMask() const {
    GpuSection section;
    if (pc_mesh_set_mask(SharedGpuContext(), gpu_, mesh_.masked_triangles.data(),
                         static_cast<int>(mesh_.masked_triangles.size())) != PC_OK)
        throw std::runtime_error(std::string("pc_mesh_set_mask: ") + pc_last_error());
}

void MakeRayCamera(const SceneTransformations& st, pc_ray_camera* out) {
    Mat4f inv;
    if (!Inverse4(MatMul4(st.view_matrix, st.model_matrix), &inv)) throw std::runtime_error("view * model is singular");
    pc_ray_camera& cam = *out;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) cam.dir_matrix[3 * r + c] = inv[4 * r + c];
        cam.origin[r] = inv[4 * r + 3];
    }
    cam.fx = st.intrinsics.fx;
    cam.fy = st.intrinsics.fy;
    cam.cx = st.intrinsics.cx;
    cam.cy = st.intrinsics.cy;
    cam.unproject_sign = st.intrinsics.convention == CameraConvention::OpenCV ? 1.0f : -1.0f;
}

void AcceleratedMesh::RayCastPixels(const SceneTransformations& st, const float* xy, size_t n, bool check_mask,
                                    std::vector<std::optional<RayHit>>& hits, bool exhaustive) const {
    hits.assign(n, std::nullopt);
    if (n == 0) return;
    GpuSection section;   // also guards the mutable host scratch below (hit_, pos_, uvt_, prim_)
    pc_context* ctx = SharedGpuContext();
    pc_ray_camera cam;
    MakeRayCamera(st, &cam);
    // the mask can be edited through inner_mut(): always send the current bits
    if (check_mask) SyncMask();
    hit_.resize(n);
    pos_.resize(3 * n);
    uvt_.resize(3 * n);
    prim_.resize(n);
    if ((exhaustive ? pc_raycast_pixels_sweep : pc_raycast_pixels)(ctx, gpu_, &cam, xy, static_cast<int>(n), check_mask ? 1 : 0,
                                                                   hit_.data(), pos_.data(), prim_.data(), uvt_.data()) != PC_OK)
        throw std::runtime_error(std::string("pc_raycast_pixels: ") + pc_last_error());
    for (size_t i = 0; i < n; i++) {
        if (!hit_[i]) continue;
        RayHit h;
        h.pos = {pos_[3 * i], pos_[3 * i + 1], pos_[3 * i + 2]};
        h.barycentric_coordinate = {uvt_[3 * i], uvt_[3 * i + 1]};
        h.t = uvt_[3 * i + 2];
        h.primitive_id = prim_[i];
        const uint32_t* tri = &mesh_.triangles[3 * static_cast<size_t>(h.primitive_id)];
        auto vtx = [&](uint32_t v) { return Vec3f{mesh_.vertices[3 * v], mesh_.vertices[3 * v + 1], mesh_.vertices[3 * v + 2]}; };
        const Vec3f p1 = vtx(tri[0]);
        Vec3f ng = Cross(vtx(tri[1]) - p1, vtx(tri[2]) - p1);  // Embree Ng = (v1 - v0) x (v2 - v0)
        const float len = Norm(ng);
        h.normal = len > 0 ? ng * (1.0f / len) : ng;
        hits[i] = h;
    }
}

std::optional<RayHit> RayCast(const AcceleratedMesh& accel_mesh, const SceneTransformations& scene_transform, Vec2f pos,
                              bool check_mask) {
    std::vector<std::optional<RayHit>> hits;
    accel_mesh.RayCastPixels(scene_transform, pos.data(), 1, check_mask, hits);
    return hits[0];
}

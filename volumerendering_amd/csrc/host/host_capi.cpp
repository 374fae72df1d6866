// host_capi.cpp -- flat C wrapper over the C++ host surface (VolumeFile / OpacityTF / ColorTF / Camera /
// MiniApp scenes / Application) so that the pytest harness and bench.py can drive the same objects a C++
// application would.  Handles are opaque pointers; every function is exception-safe at the boundary.
#include <cstring>
#include <memory>
#include <string>

#include "Application.h"
#include "DatReader.h"
#include "dicom/DicomReader.h"

using namespace med;

namespace {
struct VolumeHandle {
    std::shared_ptr<VolumeFile> v;
    std::shared_ptr<VolumeFileDcm> dcm;  // set when the volume came from DicomReader
};
VolumeFile::Size sz(int nx, int ny, int nz) { return {(uint16_t)nx, (uint16_t)ny, (uint16_t)nz}; }
}  // namespace

#define VRH_TRY(expr_default, body) \
    try { body } catch (...) { return expr_default; }

extern "C" {

// ---- VolumeFile ---------------------------------------------------------------------------------------------
void* vrh_volume_from_raw16(const uint16_t* raw, int nx, int ny, int nz)
{
    VRH_TRY(nullptr, { return new VolumeHandle{std::make_shared<VolumeFile>(VolumeFile::FromRaw(raw, sz(nx, ny, nz)))}; })
}
void* vrh_volume_from_raw32(const uint32_t* raw, int nx, int ny, int nz)
{
    VRH_TRY(nullptr, { return new VolumeHandle{std::make_shared<VolumeFile>(VolumeFile::FromRaw(raw, sz(nx, ny, nz)))}; })
}
void* vrh_volume_from_vec4(const float* vec4, int nx, int ny, int nz, uint64_t max_number)
{
    VRH_TRY(nullptr, {
        size_t n = (size_t)nx * ny * nz;
        std::vector<vrm::vec4> data(n);
        std::memcpy(static_cast<void*>(data.data()), vec4, n * sizeof(vrm::vec4));
        return new VolumeHandle{std::make_shared<VolumeFile>("", sz(nx, ny, nz), FileDataType::Float, data, (size_t)max_number)};
    })
}
void* vrh_volume_from_dat(const char* path)
{
    VRH_TRY(nullptr, { return new VolumeHandle{std::make_shared<VolumeFile>(DatImpl().ReadFile(path, false))}; })
}
int vrh_dat_write(const char* path, const uint16_t* raw, int nx, int ny, int nz)
{
    VRH_TRY(0, { return DatImpl::WriteFile(path, raw, (uint16_t)nx, (uint16_t)ny, (uint16_t)nz) ? 1 : 0; })
}
// DicomReader::ReadVolumeFile; on failure returns NULL and writes the message to err
void* vrh_volume_from_dicom(const char* path, char* err, int errlen)
{
    try {
        auto d = DicomReader::ReadVolumeFile(path);
        return new VolumeHandle{d, d};
    } catch (const std::exception& e) {
        if (err && errlen > 0) {
            std::strncpy(err, e.what(), (size_t)errlen - 1);
            err[errlen - 1] = 0;
        }
        return nullptr;
    }
}
// out: modality, X, Y, Z, BitsStored, BitsAllocated, Largest, Smallest, SliceThickness, pos[3], orient[6], spacing[2]
int vrh_dicom_params(void* h, double* out, char* main_axis, char* frame_of_reference, int len)
{
    auto* vh = static_cast<VolumeHandle*>(h);
    if (!vh->dcm) return 0;
    const DicomVolumeParams p = vh->dcm->GetVolumeParams();
    int i = 0;
    out[i++] = (double)(int)p.Modality; out[i++] = p.X; out[i++] = p.Y; out[i++] = p.Z;
    out[i++] = p.BitsStored; out[i++] = p.BitsAllocated; out[i++] = p.LargestPixelValue; out[i++] = p.SmallestPixelValue;
    out[i++] = p.SliceThickness;
    for (double v : p.ImagePositionPatient) out[i++] = v;
    for (double v : p.ImageOrientationPatient) out[i++] = v;
    for (double v : p.PixelSpacing) out[i++] = v;
    std::strncpy(main_axis, p.MainAxis.c_str(), 7);
    main_axis[7] = 0;
    std::strncpy(frame_of_reference, p.FrameOfReference.c_str(), (size_t)len - 1);
    frame_of_reference[len - 1] = 0;
    return 1;
}
// which: 0 PixelToRCS (in[0..1]), 1 RCSToPixel, 2 RCSToVoxel
void vrh_dicom_transform(void* h, int which, const float* in, float* out)
{
    auto* d = static_cast<VolumeHandle*>(h)->dcm.get();
    if (which == 0) { vrm::vec3 r = d->PixelToRCSTransform({in[0], in[1]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
    else if (which == 1) { vrm::vec2 r = d->RCSToPixelTransform({in[0], in[1], in[2]}); out[0] = r.x; out[1] = r.y; out[2] = 0; }
    else { vrm::vec3 r = d->RCSToVoxelTransform({in[0], in[1], in[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
}
int vrh_dicom_compare(void* a, void* b, int which)
{
    auto* da = static_cast<VolumeHandle*>(a)->dcm.get();
    auto* db = static_cast<VolumeHandle*>(b)->dcm.get();
    return which == 0 ? da->CompareFrameOfReference(*db) : da->CompareOrientation(*db);
}
int vrh_dicom_modality(const char* path) { VRH_TRY(0, { return (int)DicomReader::CheckModality(path); }) }

// ---- RTSTRUCT ---------------------------------------------------------------------------------------------------
void* vrh_struct_read(const char* path)
{
    VRH_TRY(nullptr, {
        auto s = DicomReader::ReadStructFile(path);
        return s ? new std::shared_ptr<StructureFileDcm>(s) : nullptr;
    })
}
void* vrh_struct_from_contours(const char* frame_of_reference, const float* points, const int* polygon_sizes,
                               const int* polygons_per_contour, int n_contours)
{
    try {
        DicomStructParams p;
        p.FrameOfReference = frame_of_reference;
        std::vector<std::vector<std::vector<float>>> data((size_t)n_contours);
        size_t poly = 0, off = 0;
        for (int c = 0; c < n_contours; ++c)
            for (int k = 0; k < polygons_per_contour[c]; ++k) {
                const int n = polygon_sizes[poly++];
                data[(size_t)c].emplace_back(points + off, points + off + n);
                off += (size_t)n;
            }
        return new std::shared_ptr<StructureFileDcm>(std::make_shared<StructureFileDcm>("", p, data));
    } catch (...) {
        return nullptr;
    }
}
void vrh_struct_free(void* s) { delete static_cast<std::shared_ptr<StructureFileDcm>*>(s); }
int vrh_struct_contour_count(void* s) { return (int)(*static_cast<std::shared_ptr<StructureFileDcm>*>(s))->GetContourData().size(); }
int vrh_struct_polygon_count(void* s, int contour)
{
    const auto& d = (*static_cast<std::shared_ptr<StructureFileDcm>*>(s))->GetContourData();
    return contour >= 0 && (size_t)contour < d.size() ? (int)d[(size_t)contour].size() : -1;
}
int vrh_struct_polygon(void* s, int contour, int polygon, float* out, int capacity)
{
    const auto& d = (*static_cast<std::shared_ptr<StructureFileDcm>*>(s))->GetContourData();
    if (contour < 0 || (size_t)contour >= d.size() || polygon < 0 || (size_t)polygon >= d[(size_t)contour].size()) return -1;
    const auto& p = d[(size_t)contour][(size_t)polygon];
    for (int i = 0; i < capacity && (size_t)i < p.size(); ++i) out[i] = p[(size_t)i];
    return (int)p.size();
}
// text: "label\nname\nframe of reference\n" then one line "number\tname\talgorithm" per ROI; colors: rgb per ROI
int vrh_struct_info(void* s, char* text, int len, float* colors, int max_colors)
{
    VRH_TRY(-1, {
        const DicomStructParams p = (*static_cast<std::shared_ptr<StructureFileDcm>*>(s))->GetStructParams();
        std::string t = p.Label + "\n" + p.Name + "\n" + p.FrameOfReference + "\n";
        for (const auto& r : p.StructureSetROISequence) t += std::to_string(r.Number) + "\t" + r.Name + "\t" + r.AlgorithmType + "\n";
        std::strncpy(text, t.c_str(), (size_t)len - 1);
        text[len - 1] = 0;
        int n = 0;
        for (const auto& c : p.DisplayColors) {
            if (n >= max_colors) break;
            colors[3 * n] = c.x;
            colors[3 * n + 1] = c.y;
            colors[3 * n + 2] = c.z;
            ++n;
        }
        return (int)p.DisplayColors.size();
    })
}
// StructureFileDcm::Create3DMask; `volume` must come from vrh_volume_from_dicom.  NULL when the reference returns nullptr.
void* vrh_struct_create_mask(void* s, void* volume, const int ids[4], unsigned post_process)
{
    try {
        auto* vh = static_cast<VolumeHandle*>(volume);
        if (!vh->dcm) return nullptr;
        const std::array<int, 4> four{ids[0], ids[1], ids[2], ids[3]};
        auto m = (*static_cast<std::shared_ptr<StructureFileDcm>*>(s))->Create3DMask(*vh->dcm, four, static_cast<ContourPostProcess>(post_process));
        if (!m) return nullptr;
        return new VolumeHandle{m, m};
    } catch (...) {
        return nullptr;
    }
}
void vrh_volume_free(void* h) { delete static_cast<VolumeHandle*>(h); }
void vrh_volume_normalize(void* h, int value) { static_cast<VolumeHandle*>(h)->v->NormalizeData(value); }
void vrh_volume_gradient(void* h, int norm01) { static_cast<VolumeHandle*>(h)->v->PreComputeGradient(norm01 != 0); }
void vrh_volume_average_gradient(void* h, int k) { static_cast<VolumeHandle*>(h)->v->AverageGradient(k); }
const float* vrh_volume_data(void* h) { return static_cast<const float*>(static_cast<VolumeHandle*>(h)->v->GetVoidPtr()); }
uint64_t vrh_volume_max_number(void* h) { return static_cast<VolumeHandle*>(h)->v->GetMaxNumber(); }
uint64_t vrh_volume_data_range(void* h) { return static_cast<VolumeHandle*>(h)->v->GetDataRange(); }
int vrh_volume_is_normalized(void* h) { return static_cast<VolumeHandle*>(h)->v->IsNormalized() ? 1 : 0; }
int vrh_volume_index(void* h, int x, int y, int z) { return static_cast<VolumeHandle*>(h)->v->GetIndexFrom3D(x, y, z); }
void vrh_volume_voxel(void* h, int x, int y, int z, float out[4])
{
    vrm::vec4 v = static_cast<VolumeHandle*>(h)->v->GetVoxelData(x, y, z);
    std::memcpy(out, &v.x, 16);
}
void vrh_volume_size(void* h, int out[3])
{
    auto [x, y, z] = static_cast<VolumeHandle*>(h)->v->GetSize();
    out[0] = x; out[1] = y; out[2] = z;
}
void vrh_volume_bbox(void* h, float out[3])
{
    auto [x, y, z] = static_cast<VolumeHandle*>(h)->v->GetBBOXSize();
    out[0] = x; out[1] = y; out[2] = z;
}
void vrh_set_worker_threads(unsigned n) { VolumeFile::SetWorkerThreads(n); }

// ---- transfer functions -------------------------------------------------------------------------------------
void* vrh_otf_create(int res) { VRH_TRY(nullptr, { return new OpacityTF(res); }) }
void vrh_otf_free(void* t) { delete static_cast<OpacityTF*>(t); }
int vrh_otf_resolution(void* t) { return static_cast<OpacityTF*>(t)->GetTextureResolution(); }
const float* vrh_otf_data(void* t) { return static_cast<OpacityTF*>(t)->GetYPoints().data(); }
void vrh_otf_reset(void* t) { static_cast<OpacityTF*>(t)->ResetTF(); }
int vrh_otf_add_cp(void* t, double x, double y) { return static_cast<OpacityTF*>(t)->AddControlPoint(x, y); }
void vrh_otf_set_cp(void* t, int id, double x, double y) { static_cast<OpacityTF*>(t)->SetControlPoint(id, x, y); }
int vrh_otf_cp_count(void* t) { return (int)static_cast<OpacityTF*>(t)->GetControlPoints().size(); }
void vrh_otf_cp(void* t, int i, double out[2])
{
    const auto& c = static_cast<OpacityTF*>(t)->GetControlPoints()[i];
    out[0] = c.x; out[1] = c.y;
}
void vrh_otf_set_data_range(void* t, int r) { static_cast<OpacityTF*>(t)->SetDataRange(r); }
int vrh_otf_data_range(void* t) { return static_cast<OpacityTF*>(t)->GetDataRange(); }
int vrh_otf_save(void* t, const char* path) { VRH_TRY(0, { return static_cast<OpacityTF*>(t)->Save(path) ? 1 : 0; }) }
void vrh_otf_load(void* t, const char* path, int rescale)
{
    try { static_cast<OpacityTF*>(t)->Load(path, rescale ? TFLoadOption::RESCALE_TO_NEW_RANGE : TFLoadOption::NONE); } catch (...) {}
}
void vrh_otf_calibrate(void* t, void* mask, void* file, const int active[4])
{
    try {
        static_cast<OpacityTF*>(t)->CalibrateOnMask(static_cast<VolumeHandle*>(mask)->v, static_cast<VolumeHandle*>(file)->v,
                                                    {active[0], active[1], active[2], active[3]});
    } catch (...) {}
}
void vrh_otf_histogram(void* t, void* file, float* out)
{
    auto* tf = static_cast<OpacityTF*>(t);
    tf->ActivateHistogram(*static_cast<VolumeHandle*>(file)->v);
    std::memcpy(out, tf->GetHistogram().data(), tf->GetHistogram().size() * sizeof(float));
}
void vrh_otf_remap_cp(void* t, double x, double y, int data_range, int tf_res, double out[2])
{
    auto r = static_cast<OpacityTF*>(t)->RemapCP({x, y}, data_range, tf_res);
    out[0] = r.x; out[1] = r.y;
}

void* vrh_ctf_create(int res) { VRH_TRY(nullptr, { return new ColorTF(res); }) }
void vrh_ctf_free(void* t) { delete static_cast<ColorTF*>(t); }
int vrh_ctf_resolution(void* t) { return static_cast<ColorTF*>(t)->GetTextureResolution(); }
const float* vrh_ctf_data(void* t) { return &static_cast<ColorTF*>(t)->GetColors()[0].x; }
void vrh_ctf_reset(void* t) { static_cast<ColorTF*>(t)->ResetTF(); }
int vrh_ctf_add_cp(void* t, double x, const float rgba[4])
{
    return static_cast<ColorTF*>(t)->AddColorControlPoint(x, vrm::vec4(rgba[0], rgba[1], rgba[2], rgba[3]));
}
void vrh_ctf_set_color(void* t, int id, const float rgba[4])
{
    static_cast<ColorTF*>(t)->SetControlColor(id, vrm::vec4(rgba[0], rgba[1], rgba[2], rgba[3]));
}
int vrh_ctf_save(void* t, const char* path) { VRH_TRY(0, { return static_cast<ColorTF*>(t)->Save(path) ? 1 : 0; }) }
void vrh_ctf_load(void* t, const char* path) { try { static_cast<ColorTF*>(t)->Load(path); } catch (...) {} }

// ---- camera ---------------------------------------------------------------------------------------------------
void* vrh_camera_create(float fov, float aspect, float n, float f)
{
    VRH_TRY(nullptr, { return new Camera(Camera::CreatePerspective(fov, aspect, n, f)); })
}
void vrh_camera_free(void* c) { delete static_cast<Camera*>(c); }
void vrh_camera_set_orbit(void* c, float pitch, float yaw, float dist) { static_cast<Camera*>(c)->SetOrbit(pitch, yaw, dist); }
void vrh_camera_rotate(void* c, float dx, float dy) { static_cast<Camera*>(c)->Rotate(dx, dy); }
void vrh_camera_zoom(void* c, float delta) { static_cast<Camera*>(c)->SetZoomDistance(delta); }
void vrh_camera_set_position(void* c, float x, float y, float z) { static_cast<Camera*>(c)->SetPosition(vrm::vec3(x, y, z)); }
void vrh_camera_key(void* c, int key) { static_cast<Camera*>(c)->KeyboardEvent(key); }
// out: view[16], proj[16], view_inv[16], proj_inv[16], position[3], forward[3]
void vrh_camera_get(void* c, float* out)
{
    auto* cam = static_cast<Camera*>(c);
    std::memcpy(out, cam->GetViewMatrix().data(), 64);
    std::memcpy(out + 16, cam->GetProjectionMatrix().data(), 64);
    std::memcpy(out + 32, cam->GetInverseViewMatrix().data(), 64);
    std::memcpy(out + 48, cam->GetInverseProjectionMatrix().data(), 64);
    vrm::vec3 p = cam->GetPosition(), f = cam->GetForward();
    out[64] = p.x; out[65] = p.y; out[66] = p.z;
    out[67] = f.x; out[68] = f.y; out[69] = f.z;
}

// ---- Application + scenes ---------------------------------------------------------------------------------
void* vrh_app_create(uint32_t w, uint32_t h, int device) { VRH_TRY(nullptr, { return new Application(w, h, device); }) }
void vrh_app_free(void* a) { delete static_cast<Application*>(a); }
int vrh_app_ok(void* a) { return static_cast<Application*>(a)->Ok() ? 1 : 0; }
const char* vrh_app_error(void* a) { return static_cast<Application*>(a)->LastError().c_str(); }
void* vrh_app_context(void* a) { return static_cast<Application*>(a)->GetContext(); }
void* vrh_app_camera(void* a) { return &static_cast<Application*>(a)->GetCamera(); }

// variant: vr_variant; volumes in the slot order of include/vr.h (TF_CALIB: ct, filled mask, un-filled mask)
int vrh_app_start(void* a, int variant, void* v0, void* v1, void* v2, int tf_res)
{
    try {
        auto* app = static_cast<Application*>(a);
        auto vol = [](void* h) { return h ? static_cast<VolumeHandle*>(h)->v : nullptr; };
        std::unique_ptr<MiniApp> scene;
        switch (variant) {
        case VR_VARIANT_BASIC: scene = std::make_unique<BasicVolumeApp>(vol(v0), tf_res > 0 ? tf_res : 256); break;
        case VR_VARIANT_LIGHT: scene = std::make_unique<BasicVolLightApp>(vol(v0), tf_res > 0 ? tf_res : 4096); break;
        case VR_VARIANT_LIGHT_INSHADER: {
            auto m = std::make_unique<BasicVolLightApp>(vol(v0), tf_res > 0 ? tf_res : 4096);
            m->SetInShaderGradient(true);
            scene = std::move(m);
            break;
        }
        case VR_VARIANT_VOLUME_MASK: scene = std::make_unique<VolumeMaskApp>(vol(v0), vol(v1), vol(v2)); break;
        case VR_VARIANT_THREE_FILES: scene = std::make_unique<ThreeFilesApp>(vol(v0), vol(v1), vol(v2)); break;
        case VR_VARIANT_MULTI_CTRT: scene = std::make_unique<MultiCTRTApp>(vol(v0), vol(v1)); break;
        case VR_VARIANT_ILLUSTRATIVE: {
            auto m = std::make_unique<MultiCTRTApp>(vol(v0), vol(v1));
            m->SetIllustrative(true);
            scene = std::move(m);
            break;
        }
        case VR_VARIANT_TF_CALIB: scene = std::make_unique<TFCalibrationApp>(vol(v0), vol(v1), vol(v2)); break;
        default: return VR_ERR_INVALID_ARG;
        }
        return app->OnStart(std::move(scene));
    } catch (...) {
        return VR_ERR_INVALID_ARG;
    }
}
void vrh_app_set_prepare_on_device(void* a, int on) { static_cast<Application*>(a)->m_PrepareOnDevice = on != 0; }
int vrh_app_update(void* a) { VRH_TRY(VR_ERR_HIP, { return static_cast<Application*>(a)->OnUpdate(); }) }
int vrh_app_render(void* a) { VRH_TRY(VR_ERR_HIP, { return static_cast<Application*>(a)->OnRender(); }) }
int vrh_app_resize(void* a, uint32_t w, uint32_t h) { VRH_TRY(VR_ERR_HIP, { return static_cast<Application*>(a)->OnResize(w, h); }) }
int vrh_app_read_frame(void* a, float* frag, uint8_t* bgra, uint64_t* samples)
{
    VRH_TRY(VR_ERR_HIP, { return static_cast<Application*>(a)->ReadFrame(frag, bgra, samples); })
}
void vrh_app_set_params(void* a, int fragment_mode, int steps_count, float step_size, const float clips[6], const int toggles[4])
{
    auto* app = static_cast<Application*>(a);
    app->m_FragmentMode = fragment_mode;
    if (steps_count >= 0) app->m_StepsCount = steps_count;
    if (step_size > 0.0f) app->m_StepSize = step_size;
    if (clips) {
        app->m_ClipsX = {clips[0], clips[1]};
        app->m_ClipsY = {clips[2], clips[3]};
        app->m_ClipsZ = {clips[4], clips[5]};
    }
    if (toggles)
        for (int i = 0; i < 4; ++i) app->m_BToggles[i] = toggles[i] != 0;
}
void vrh_app_get_stepping(void* a, int* steps_count, float* step_size)
{
    auto* app = static_cast<Application*>(a);
    *steps_count = app->m_StepsCount;
    *step_size = app->m_StepSize;
}
void vrh_app_get_uniforms(void* a, vr_uniforms* out) { *out = static_cast<Application*>(a)->GetUniforms(); }
// the TF objects of the running scene: which = 0 CT/only pair, 1 RT pair; returns OpacityTF* / ColorTF*
void* vrh_app_scene_otf(void* a, int which)
{
    MiniApp* s = static_cast<Application*>(a)->GetApp();
    if (auto* p = dynamic_cast<BasicVolumeApp*>(s)) return p->p_OpacityTf.get();
    if (auto* p = dynamic_cast<BasicVolLightApp*>(s)) return p->p_OpacityTf.get();
    if (auto* p = dynamic_cast<VolumeMaskApp*>(s)) return which ? p->p_OpacityTfRT.get() : p->p_OpacityTfCT.get();
    if (auto* p = dynamic_cast<ThreeFilesApp*>(s)) return which ? p->p_OpacityTfRT.get() : p->p_OpacityTfCT.get();
    if (auto* p = dynamic_cast<MultiCTRTApp*>(s)) return which ? p->p_OpacityTfRT.get() : p->p_OpacityTfCT.get();
    if (auto* p = dynamic_cast<TFCalibrationApp*>(s)) return p->p_OpacityTfCT.get();
    return nullptr;
}
void* vrh_app_scene_ctf(void* a, int which)
{
    MiniApp* s = static_cast<Application*>(a)->GetApp();
    if (auto* p = dynamic_cast<BasicVolumeApp*>(s)) return p->p_ColorTf.get();
    if (auto* p = dynamic_cast<BasicVolLightApp*>(s)) return p->p_ColorTf.get();
    if (auto* p = dynamic_cast<VolumeMaskApp*>(s)) return which ? p->p_ColorTfRT.get() : p->p_ColorTfCT.get();
    if (auto* p = dynamic_cast<ThreeFilesApp*>(s)) return which ? p->p_ColorTfRT.get() : p->p_ColorTfCT.get();
    if (auto* p = dynamic_cast<MultiCTRTApp*>(s)) return which ? p->p_ColorTfRT.get() : p->p_ColorTfCT.get();
    if (auto* p = dynamic_cast<TFCalibrationApp*>(s)) return p->p_ColorTfCT.get();
    return nullptr;
}

}  // extern "C"

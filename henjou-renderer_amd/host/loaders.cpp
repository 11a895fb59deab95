// Scene surface: render_option.json and glTF 2.0 loaders + per-frame animation / camera evaluation.
// Restates loader/render_json_loader.h:14-228, loader/gltfloader.h:1068-1601, renderer/renderer.h:257-291,1145-1169
// on top of the local JSON parser (the reference's nlohmann/json, tinygltf and glm are un-vendored submodules).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <sstream>

#include "json.hpp"
#include "scene.hpp"

namespace hjr {

bool read_png_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);
bool read_image_rgba8(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h, std::string& err);

static bool read_file(const std::string& path, std::string& out)
{
    std::ifstream ifs(path, std::ios::binary);
    if (ifs.fail()) return false;
    std::ostringstream ss;
    ss << ifs.rdbuf();
    out = ss.str();
    return true;
}
static void put_str(char* dst, size_t cap, const std::string& s)
{
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(dst, s.data(), n);
    dst[n] = 0;
}

// fpsLoader — render_json_loader.h:14-34: last parsable line of ./fps.txt wins
static bool fps_loader(unsigned int& fps, const std::string& path)
{
    std::ifstream ifs(path);
    if (ifs.fail()) return false;
    std::string str;
    bool any = false;
    try {
        while (std::getline(ifs, str)) { fps = (unsigned int)std::stoi(str); any = true; }
    } catch (std::exception&) { return false; }
    return any;
}

bool load_render_option(const std::string& path, hjr_render_option& o, std::string& err)
{
    memset(&o, 0, sizeof(o));
    o.camera_animation_id = -1;
    o.seed = 1;
    o.integrator = HJR_INTEGRATOR_NEE;
    o.devices = 1;
    o.tile = 8;
    std::string text;
    if (!read_file(path, text)) { err = "File " + path + " not found"; return false; }
    try {
        Json j = Json::parse(text);
        const Json& img = j.at("Image");
        o.image_width = (uint32_t)img.at("image_width").as_number();
        o.image_height = (uint32_t)img.at("image_height").as_number();
        put_str(o.image_name, sizeof(o.image_name), img.at("image_name").as_string());
        put_str(o.image_directory, sizeof(o.image_directory), img.at("image_directory").as_string());
        o.max_spp = (uint32_t)img.at("max_spp").as_number();

        const Json& g = j.at("GLTF_file");
        put_str(o.gltf_path, sizeof(o.gltf_path), g.at("gltf_filepath").as_string());
        put_str(o.gltf_name, sizeof(o.gltf_name), g.at("gltf_filename").as_string());

        const std::string& mode = j.at("Render_mode").as_string(); // render_json_loader.h:116-136: unknown -> Default
        if (mode == "Denoise") o.render_mode = HJR_MODE_DENOISE;
        else if (mode == "Debug") o.render_mode = HJR_MODE_DEBUG;
        else if (mode == "DenoiseUpScale2X") o.render_mode = HJR_MODE_DENOISE_UPSCALE2X;
        else o.render_mode = HJR_MODE_DEFAULT;

        const Json& cam = j.at("Camera");
        for (int k = 0; k < 3; k++) {
            o.camera_position[k] = (float)cam.at("camera_position").at(k).as_number();
            o.camera_direction[k] = (float)cam.at("camera_direction").at(k).as_number();
        }
        o.camera_fov = (float)(M_PI * cam.at("camera_fov").as_number() / 180.0f); // :144, degrees -> radians in double
        o.allow_camera_animation = cam.at("allow_camera_animation").as_bool() ? 1 : 0;

        put_str(o.ptxfile_path, sizeof(o.ptxfile_path), j.at("PTX_File").at("ptxfile_path").as_string());

        const Json& an = j.at("Animation");
        o.fps = (uint32_t)an.at("fps").as_number();
        o.start_frame = (uint32_t)an.at("start_frame").as_number();
        o.end_frame = (uint32_t)an.at("end_frame").as_number();
        o.time_limit = (float)an.at("time_limit").as_number();
        unsigned int loaded_fps;
        if (fps_loader(loaded_fps, "./fps.txt")) o.fps = loaded_fps; // :164-171

        const Json& sky = j.at("Sky");
        put_str(o.IBL_path, sizeof(o.IBL_path), sky.at("IBL_path").as_string());
        o.IBL_intensity = (float)sky.at("IBL_intensity").as_number();
        o.use_IBL = sky.at("use_IBL").as_bool() ? 1 : 0;
        for (int k = 0; k < 3; k++) o.scene_sky_default[k] = (float)sky.at("scene_sky_default").at(k).as_number();

        const Json& op = j.at("Option");
        o.use_date = op.at("use_date").as_bool() ? 1 : 0;
        o.save_renderOption = op.at("save_renderOption").as_bool() ? 1 : 0;

        put_str(o.LUT_path, sizeof(o.LUT_path), j.at("LUT").at("LUT_path").as_string());

        if (o.save_renderOption) { // render_json_loader.h:204-219: timestamped copy of the JSON text in the CWD
            std::time_t now = std::time(nullptr);
            std::string stamp = std::ctime(&now);
            stamp.erase(std::remove(stamp.begin(), stamp.end(), ':'), stamp.end());
            stamp.erase(std::remove(stamp.begin(), stamp.end(), '\n'), stamp.end());
            std::ofstream file("renderoption" + stamp + ".json");
            file << text;
        }

        if (const Json* h = j.find("Henjou_HIP")) { // extension section; the reference never reads it
            o.seed = (uint32_t)h->number_or("seed", 1);
            std::string in = h->string_or("integrator", "NEE");
            o.integrator = in == "MIS" ? HJR_INTEGRATOR_MIS : (in == "Pathtrace" ? HJR_INTEGRATOR_PT : HJR_INTEGRATOR_NEE);
            // devices: how many GPUs of the node share a frame (8x8 pixel tiles dealt round-robin, one process per GPU);
            // tile: the shard granularity, fixed at 8 (one wavefront of pixels) — any other value is rejected rather than ignored
            const double dv = h->number_or("devices", 1), tl = h->number_or("tile", 8);
            if (!(dv >= 1 && dv <= 64) || dv != (double)(uint32_t)dv) throw JsonError("Henjou_HIP.devices must be an integer in [1, 64]");
            if (tl != 8) throw JsonError("Henjou_HIP.tile: only 8 (8x8 pixel tiles) is supported");
            o.devices = (uint32_t)dv;
            o.tile = 8;
            // serial_io: hjr_render_file without the overlap of the output stage / the next frame's host preparation with the render;
            // fast_math: the approximate-arithmetic kernels (HJR_FLAG_FAST_MATH; pictures within the metric's RMSE tolerance, not bit-exact)
            auto flag = [&](const char* k) { const Json* v = h->find(k); return v ? (v->is_bool() ? v->as_bool() : v->as_number() != 0.0) : false; };
            o.serial_io = flag("serial_io") ? 1 : 0;
            o.fast_math = flag("fast_math") ? 1 : 0;
            o.force_rebuild = flag("force_rebuild") ? 1 : 0;
        }
    } catch (std::exception& e) { // :222-225
        err = std::string("Caught exception: ") + e.what();
        return false;
    }
    return true;
}

// ---------------------------------------------------------------- glTF
namespace {

struct Buffer { std::string data; };
struct BufferView { int buffer = 0; size_t byteOffset = 0, byteLength = 0, byteStride = 0; };
struct Accessor { int bufferView = -1; size_t byteOffset = 0; int componentType = 0; size_t count = 0; std::string type; };

int comp_size(int ct)
{
    switch (ct) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    case 5130: return 8;
    default: return 0;
    }
}
int type_comps(const std::string& t)
{
    if (t == "SCALAR") return 1;
    if (t == "VEC2") return 2;
    if (t == "VEC3") return 3;
    if (t == "VEC4") return 4;
    if (t == "MAT2") return 4;
    if (t == "MAT3") return 9;
    if (t == "MAT4") return 16;
    return 0;
}

int b64val(char c)
{
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}
std::string b64decode(const std::string& s, size_t from)
{
    std::string o;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); i++) {
        int v = b64val(s[i]);
        if (v < 0) continue;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; o += (char)((acc >> bits) & 0xFF); }
    }
    return o;
}

struct Model {
    Json json;
    std::vector<Buffer> buffers;
    std::vector<BufferView> views;
    std::vector<Accessor> accessors;

    // (pointer, stride, count) of an accessor, bounds-checked — arrayAdapter, gltfloader.h:925-951
    void span(int acc, const unsigned char*& ptr, size_t& stride, size_t& count, size_t elem_bytes) const
    {
        if (acc < 0 || (size_t)acc >= accessors.size()) throw JsonError("accessor index out of range");
        const Accessor& a = accessors[acc];
        if (a.bufferView < 0 || (size_t)a.bufferView >= views.size()) throw JsonError("accessor without bufferView");
        const BufferView& v = views[a.bufferView];
        if (v.buffer < 0 || (size_t)v.buffer >= buffers.size()) throw JsonError("bufferView.buffer out of range");
        const std::string& d = buffers[v.buffer].data;
        // Accessor::ByteStride(bufferView): tight packing when byteStride is 0
        size_t tight = (size_t)comp_size(a.componentType) * (size_t)type_comps(a.type);
        stride = v.byteStride ? v.byteStride : tight;
        count = a.count;
        size_t off = v.byteOffset + a.byteOffset;
        if (count && (off + (count - 1) * stride + elem_bytes > d.size())) throw JsonError("accessor reads past the end of its buffer");
        ptr = reinterpret_cast<const unsigned char*>(d.data()) + off;
    }
};

template <typename T> T rd(const unsigned char* p)
{
    T v;
    memcpy(&v, p, sizeof(T));
    return v;
}

void parse_model(const std::string& dir, const std::string& file, Model& m)
{
    std::string path = dir + "/" + file;
    std::string raw;
    if (!read_file(path, raw)) throw JsonError("Failed to parse glTF: cannot open " + path);
    std::string ext;
    size_t dot = file.find_last_of('.');
    if (dot != std::string::npos) ext = file.substr(dot + 1);
    std::string glb_bin;
    bool have_glb_bin = false;
    if (ext == "glb") { // gltfloader.h:1086-1091
        if (raw.size() < 20 || raw.compare(0, 4, "glTF") != 0) throw JsonError("bad GLB header");
        uint32_t total = rd<uint32_t>((const unsigned char*)raw.data() + 8);
        if (total > raw.size()) throw JsonError("truncated GLB");
        size_t pos = 12;
        std::string jtxt;
        while (pos + 8 <= total) {
            uint32_t clen = rd<uint32_t>((const unsigned char*)raw.data() + pos);
            uint32_t ctype = rd<uint32_t>((const unsigned char*)raw.data() + pos + 4);
            pos += 8;
            if (pos + clen > total) throw JsonError("truncated GLB chunk");
            if (ctype == 0x4E4F534A) jtxt = raw.substr(pos, clen);
            else if (ctype == 0x004E4942 && !have_glb_bin) { glb_bin = raw.substr(pos, clen); have_glb_bin = true; }
            pos += clen;
        }
        m.json = Json::parse(jtxt);
    } else {
        m.json = Json::parse(raw);
    }
    const Json& j = m.json;
    if (const Json* bs = j.find("buffers")) {
        for (size_t i = 0; i < bs->size(); i++) {
            const Json& b = bs->at(i);
            Buffer buf;
            const Json* uri = b.find("uri");
            if (!uri) {
                if (i == 0 && have_glb_bin) buf.data = glb_bin;
                else throw JsonError("buffer without uri");
            } else {
                const std::string& u = uri->as_string();
                if (u.compare(0, 5, "data:") == 0) {
                    size_t c = u.find("base64,");
                    if (c == std::string::npos) throw JsonError("unsupported data URI");
                    buf.data = b64decode(u, c + 7);
                } else if (!read_file(dir + "/" + u, buf.data)) throw JsonError("cannot open buffer file " + dir + "/" + u);
            }
            size_t want = (size_t)b.number_or("byteLength", 0);
            if (buf.data.size() < want) throw JsonError("buffer shorter than its byteLength");
            m.buffers.push_back(std::move(buf));
        }
    }
    if (const Json* vs = j.find("bufferViews")) {
        for (size_t i = 0; i < vs->size(); i++) {
            const Json& v = vs->at(i);
            BufferView bv;
            bv.buffer = (int)v.int_or("buffer", 0);
            bv.byteOffset = (size_t)v.int_or("byteOffset", 0);
            bv.byteLength = (size_t)v.int_or("byteLength", 0);
            bv.byteStride = (size_t)v.int_or("byteStride", 0);
            m.views.push_back(bv);
        }
    }
    if (const Json* as = j.find("accessors")) {
        for (size_t i = 0; i < as->size(); i++) {
            const Json& a = as->at(i);
            Accessor ac;
            ac.bufferView = (int)a.int_or("bufferView", -1);
            ac.byteOffset = (size_t)a.int_or("byteOffset", 0);
            ac.componentType = (int)a.int_or("componentType", 0);
            ac.count = (size_t)a.int_or("count", 0);
            ac.type = a.string_or("type", "");
            m.accessors.push_back(ac);
        }
    }
}

double arr_or(const Json* a, size_t i, double d)
{
    return (a && a->is_array() && i < a->size() && a->at(i).is_number()) ? a->at(i).num : d;
}

// loadTexture(textures, known_tex, name, modelpath, type) — texture_load.h:7-20: de-duplicate by file name, return slot.
// Texture(filename, type) (renderer/texture.h:22-38) decodes with stbi_load(..., STBI_rgb_alpha); here: the PNG decoder of
// image_io.cpp or the baseline JPEG decoder of jpeg.cpp, by file signature (other formats and progressive JPEG are reported
// as an error instead of the reference's silent "NOT FOUND").
int texture_slot(SceneData& sc, std::map<std::string, int>& known, const Model& m, const Json* texinfo, const std::string& dir, bool srgb)
{
    if (!texinfo) return -1;
    int idx = (int)texinfo->int_or("index", -1);
    if (idx < 0) return -1;
    const Json& tex = m.json.at("textures").at((size_t)idx);
    int src = (int)tex.int_or("source", -1);
    if (src < 0) return -1;
    std::string uri = m.json.at("images").at((size_t)src).string_or("uri", "");
    auto it = known.find(uri);
    if (it != known.end()) return it->second;
    if (uri == "") return -1;
    int slot = (int)sc.texture_files.size();
    SceneData::TexturePixels px;
    int w = 0, h = 0;
    std::string err;
    if (!read_image_rgba8(dir + "/" + uri, px.rgba, w, h, err)) throw JsonError("texture " + uri + ": " + err);
    px.width = (uint32_t)w; px.height = (uint32_t)h; px.srgb = srgb ? 1 : 0;
    sc.texture_files.push_back(uri);
    sc.textures.push_back(std::move(px));
    known[uri] = slot;
    return slot;
}

} // namespace

bool load_gltf(const std::string& dir, const std::string& file, SceneData& sc, hjr_render_option& opt, std::string& err)
{
    try {
        Model m;
        parse_model(dir, file, m);
        const Json& j = m.json;
        const Json* nodes = j.find("nodes");
        size_t n_nodes = nodes ? nodes->size() : 0;
        std::vector<NodeMotion> animation(n_nodes); // gltfloader.h:1120-1121

        // ---- materials (gltfloader.h:1125-1267).  tinygltf defaults: baseColorFactor 1, metallic 1, roughness 1, emissive 0.
        std::map<std::string, int> known_tex;
        if (const Json* mats = j.find("materials")) {
            for (size_t i = 0; i < mats->size(); i++) {
                const Json& material = mats->at(i);
                const Json* pbr = material.find("pbrMetallicRoughness");
                hjr_material mat;
                memset(&mat, 0, sizeof(mat));
                const Json* bcf = pbr ? pbr->find("baseColorFactor") : nullptr;
                for (int k = 0; k < 3; k++) mat.basecolor[k] = float(arr_or(bcf, (size_t)k, 1.0));
                mat.basecolor_tex = texture_slot(sc, known_tex, m, pbr ? pbr->find("baseColorTexture") : nullptr, dir, true);
                mat.roughness = float(pbr ? pbr->number_or("roughnessFactor", 1.0) : 1.0);
                mat.metallic_roughness_tex = texture_slot(sc, known_tex, m, pbr ? pbr->find("metallicRoughnessTexture") : nullptr, dir, false);
                mat.metallic = float(pbr ? pbr->number_or("metallicFactor", 1.0) : 1.0);
                const Json* em = material.find("emissiveFactor");
                for (int k = 0; k < 3; k++) mat.emission[k] = float(arr_or(em, (size_t)k, 0.0));
                mat.is_light = (mat.emission[0] + mat.emission[1] + mat.emission[2] > 0.0) ? 1 : 0; // :1162-1167 (before strength)
                mat.normal_tex = texture_slot(sc, known_tex, m, material.find("normalTexture"), dir, false);
                mat.emission_tex = -1; // gltfloader.h:1159
                mat.sheen = 0;
                mat.clearcoat = 0;
                mat.transmission = 0;
                mat.ior = 1.0;
                mat.is_thinfilm = 0;
                if (const Json* exts = material.find("extensions")) {
                    for (auto& kv : exts->obj) {
                        const Json& o = kv.second;
                        if (!o.is_object()) continue;
                        if (kv.first == "KHR_materials_clearcoat") {
                            if (const Json* v = o.find("clearcoatFactor")) mat.clearcoat = (float)v->as_number();
                        } else if (kv.first == "KHR_materials_sheen") {
                            if (const Json* v = o.find("sheenRoughnessFactor")) mat.sheen = (float)v->as_number(); // sic (:1211)
                        } else if (kv.first == "KHR_materials_transmission") {
                            if (const Json* v = o.find("transmissionFactor")) mat.transmission = (float)v->as_number();
                        } else if (kv.first == "KHR_materials_ior") {
                            if (const Json* v = o.find("ior")) mat.ior = (float)v->as_number();
                        } else if (kv.first == "KHR_materials_emissive_strength") {
                            if (const Json* v = o.find("emissiveStrength")) {
                                float s = (float)v->as_number(); // float3 *= double -> float scalar
                                for (int k = 0; k < 3; k++) mat.emission[k] *= s;
                            }
                        } else if (kv.first == "ThinFilm") {
                            if (o.find("is_ThinFilm")) mat.is_thinfilm = 1; // presence, not value (:1253-1255)
                        }
                    }
                }
                mat.ideal_specular = (mat.roughness == 0 && mat.transmission > 0) ? 1 : 0; // :1260-1263
                sc.materials.push_back(mat);
                sc.material_names.push_back(material.string_or("name", ""));
            }
        }

        // ---- nodes: TRS as key 0, mesh nodes de-indexed, camera node (gltfloader.h:1308-1531)
        for (size_t node_index = 0; node_index < n_nodes; node_index++) {
            const Json& node = nodes->at(node_index);
            NodeMotion& na = animation[node_index];
            const Json* t = node.find("translation");
            const Json* r = node.find("rotation");
            const Json* s = node.find("scale");
            na.translation.push(0, t && t->size() ? float3_{ (float)t->at(0).num, (float)t->at(1).num, (float)t->at(2).num } : float3_{ 0, 0, 0 });
            na.rotation.push(0, r && r->size() ? float4_{ (float)r->at(0).num, (float)r->at(1).num, (float)r->at(2).num, (float)r->at(3).num } : float4_{ 0, 0, 0, 1 });
            na.scale.push(0, s && s->size() ? float3_{ (float)s->at(0).num, (float)s->at(1).num, (float)s->at(2).num } : float3_{ 1, 1, 1 });

            int mesh = (int)node.int_or("mesh", -1);
            int camera = (int)node.int_or("camera", -1);
            if (mesh != -1) {
                const Json& meshj = j.at("meshes").at((size_t)mesh);
                GeometryData gas;
                gas.index_offset = (uint32_t)sc.indices.size();
                sc.prim_offset.push_back(gas.index_offset / 3);
                uint32_t prim_id = (uint32_t)(sc.vertices.size() / 3);
                const Json& prims = meshj.at("primitives");
                for (size_t pi = 0; pi < prims.size(); pi++) {
                    const Json& prim = prims.at(pi);
                    int ind_acc = (int)prim.int_or("indices", -1);
                    if (ind_acc < 0) throw JsonError("primitive without indices (undefined behaviour in the reference, gltfloader.h:1365)");
                    int material = (int)prim.int_or("material", -1);
                    if (material < 0 || (size_t)material >= sc.materials.size())
                        throw JsonError("primitive without a valid material (unguarded in the reference, gltfloader.h:1494)");
                    const unsigned char* ip; size_t istride, icount;
                    int ict = m.accessors.at((size_t)ind_acc).componentType;
                    m.span(ind_acc, ip, istride, icount, (size_t)comp_size(ict));
                    auto index_at = [&](size_t k) -> unsigned int { // intArray<T>::operator[], gltfloader.h:992-1002
                        const unsigned char* p = ip + k * istride;
                        switch (ict) {
                        case 5120: return (unsigned int)rd<int8_t>(p);
                        case 5121: return (unsigned int)rd<uint8_t>(p);
                        case 5122: return (unsigned int)rd<int16_t>(p);
                        case 5123: return (unsigned int)rd<uint16_t>(p);
                        case 5124: return (unsigned int)rd<int32_t>(p);
                        case 5125: return (unsigned int)rd<uint32_t>(p);
                        default: throw JsonError("unsupported index component type");
                        }
                    };
                    const unsigned char *vp = nullptr, *np = nullptr, *tp = nullptr;
                    size_t vstride = 0, vcount = 0, nstride = 0, ncount = 0, tstride = 0, tcount = 0;
                    const Json& attrs = prim.at("attributes");
                    if (const Json* a = attrs.find("POSITION")) m.span((int)a->as_int(), vp, vstride, vcount, 12);
                    if (const Json* a = attrs.find("NORMAL")) m.span((int)a->as_int(), np, nstride, ncount, 12);
                    if (const Json* a = attrs.find("TEXCOORD_0")) m.span((int)a->as_int(), tp, tstride, tcount, 8);
                    if (!vp) throw JsonError("primitive without POSITION");
                    for (size_t tri = 0; tri < icount / 3; tri++) {
                        unsigned int idx[3] = { index_at(tri * 3), index_at(tri * 3 + 1), index_at(tri * 3 + 2) };
                        float3_ vert[3], norm[3];
                        float2_ texc[3];
                        for (int k = 0; k < 3; k++) {
                            if (idx[k] >= vcount) throw JsonError("Tried to access beyond the last element of an array adapter");
                            const unsigned char* p = vp + idx[k] * vstride;
                            vert[k] = { rd<float>(p), rd<float>(p + 4), rd<float>(p + 8) };
                        }
                        if (np) {
                            for (int k = 0; k < 3; k++) {
                                if (idx[k] >= ncount) throw JsonError("Tried to access beyond the last element of an array adapter");
                                const unsigned char* p = np + idx[k] * nstride;
                                norm[k] = { rd<float>(p), rd<float>(p + 4), rd<float>(p + 8) };
                            }
                        } else { // :1465-1470
                            float3_ gn = normalize3(cross3(vert[1] - vert[0], vert[2] - vert[0]));
                            norm[0] = norm[1] = norm[2] = gn;
                        }
                        if (tp) {
                            for (int k = 0; k < 3; k++) {
                                if (idx[k] >= tcount) throw JsonError("Tried to access beyond the last element of an array adapter");
                                const unsigned char* p = tp + idx[k] * tstride;
                                texc[k] = { rd<float>(p), rd<float>(p + 4) };
                            }
                        } else texc[0] = texc[1] = texc[2] = { 0, 0 };
                        for (int k = 0; k < 3; k++) {
                            sc.vertices.push_back(vert[k]);
                            sc.normals.push_back(norm[k]);
                            sc.texcoords.push_back(texc[k]);
                            sc.indices.push_back((uint32_t)sc.indices.size());
                        }
                        sc.material_ids.push_back((uint32_t)material);
                        if (sc.materials[(size_t)material].is_light) { // :1496-1500
                            sc.light_prim_ids.push_back(prim_id);
                            const float* e = sc.materials[(size_t)material].emission;
                            sc.light_prim_emission.push_back({ e[0], e[1], e[2] });
                        }
                        prim_id++;
                    }
                }
                gas.index_count = (uint32_t)sc.indices.size() - gas.index_offset;
                InstanceData ias;
                ias.animation_id = (uint32_t)node_index;
                ias.geometry_id = (uint32_t)sc.geometries.size();
                sc.geometries.push_back(gas);
                sc.instances.push_back(ias);
            } else if (camera != -1 && opt.allow_camera_animation) { // :1514-1522
                opt.camera_position[0] = 0; opt.camera_position[1] = 0; opt.camera_position[2] = 0;
                opt.camera_direction[0] = 0; opt.camera_direction[1] = 0; opt.camera_direction[2] = -1;
                opt.camera_animation_id = (int32_t)node_index;
                const Json& cam = j.at("cameras").at((size_t)camera);
                opt.camera_fov = (float)cam.at("perspective").at("yfov").as_number();
            }
        }

        // ---- animation channels appended after key 0 (gltfloader.h:1536-1589); sampler picked by CHANNEL index (:1541)
        if (const Json* anims = j.find("animations")) {
            for (size_t ai = 0; ai < anims->size(); ai++) {
                const Json& anim = anims->at(ai);
                const Json& channels = anim.at("channels");
                const Json& samplers = anim.at("samplers");
                for (size_t i = 0; i < channels.size(); i++) {
                    const Json& sampler = samplers.at(i);
                    const Json& channel = channels.at(i);
                    const Json& target = channel.at("target");
                    int target_node = (int)target.int_or("node", -1);
                    if (target_node < 0 || (size_t)target_node >= n_nodes) throw JsonError("animation channel targets a missing node");
                    const std::string path = target.string_or("path", "");
                    const unsigned char *kp, *dp;
                    size_t kstride, kcount, dstride, dcount;
                    m.span((int)sampler.at("input").as_int(), kp, kstride, kcount, 4);
                    NodeMotion& a = animation[(size_t)target_node];
                    auto key_at = [&](size_t k) -> float {
                        if (k >= kcount) throw JsonError("Tried to access beyond the last element of an array adapter");
                        return rd<float>(kp + k * kstride);
                    };
                    if (path == "translation" || path == "scale") {
                        m.span((int)sampler.at("output").as_int(), dp, dstride, dcount, 12);
                        Track<float3_>& d = (path == "translation") ? a.translation : a.scale;
                        for (size_t k = 0; k < dcount; k++) {
                            const unsigned char* p = dp + k * dstride;
                            const float3_ v = { rd<float>(p), rd<float>(p + 4), rd<float>(p + 8) }; // the value is read before its key, as in the reference
                            d.push(key_at(k), v);
                        }
                    } else if (path == "rotation") {
                        m.span((int)sampler.at("output").as_int(), dp, dstride, dcount, 16);
                        for (size_t k = 0; k < dcount; k++) {
                            const unsigned char* p = dp + k * dstride;
                            const float4_ v = { rd<float>(p), rd<float>(p + 4), rd<float>(p + 8), rd<float>(p + 12) };
                            a.rotation.push(key_at(k), v);
                        }
                    }
                }
            }
        }
        sc.animations = animation;
        for (auto& t : sc.textures) sc.texture_views.push_back(hjr_texture{ t.rgba.data(), t.width, t.height, t.srgb, 0 });
        for (size_t i = 0; i < sc.instances.size(); i++) {
            sc.geo_index_offset.push_back(sc.geometries[sc.instances[i].geometry_id].index_offset);
            sc.geo_index_count.push_back(sc.geometries[sc.instances[i].geometry_id].index_count);
            sc.inst_animation_id.push_back(sc.instances[i].animation_id);
        }
    } catch (std::exception& e) {
        err = std::string("Failed to parse glTF: ") + e.what();
        return false;
    }
    return true;
}

// ---------------------------------------------------------------- per-frame evaluation
void affine_inverse_3x4(const float* m, float* inv)
{
    // inverse of [A t; 0 1]: A^-1 by cofactors in double, t' = -A^-1 t; rounded once to float.
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double id = 1.0 / det;
    double r[9] = { A * id, -(b * i - c * h) * id, (b * f - c * e) * id,
                    B * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                    C * id, -(a * h - b * g) * id, (a * e - b * d) * id };
    double tx = m[3], ty = m[7], tz = m[11];
    for (int row = 0; row < 3; row++) {
        inv[row * 4 + 0] = (float)r[row * 3 + 0];
        inv[row * 4 + 1] = (float)r[row * 3 + 1];
        inv[row * 4 + 2] = (float)r[row * 3 + 2];
        inv[row * 4 + 3] = (float)(-(r[row * 3 + 0] * tx + r[row * 3 + 1] * ty + r[row * 3 + 2] * tz));
    }
}

void eval_transforms(const SceneData& sc, float time, float* m12, float* inv12) // renderer.h:257-291
{
    for (size_t i = 0; i < sc.instances.size(); i++) {
        sc.animations[sc.instances[i].animation_id].matrix3x4(time, m12 + i * 12);
        affine_inverse_3x4(m12 + i * 12, inv12 + i * 12);
    }
}

void eval_camera(const SceneData& sc, const hjr_render_option& o, float time, hjr_camera& cam) // renderer.h:1145-1169
{
    cam.f = (float)(2.0 / std::tan(o.camera_fov)); // full-angle fov, sic (:1147)
    float3_ cpos = { o.camera_position[0], o.camera_position[1], o.camera_position[2] };
    float3_ cdir = { o.camera_direction[0], o.camera_direction[1], o.camera_direction[2] };
    float3_ pos, dir, up, right;
    if (o.camera_animation_id != -1 && o.allow_camera_animation && (size_t)o.camera_animation_id < sc.animations.size()) {
        // camera node: position through the node's T*R*S, direction and up through its rotation alone (w = 0); the products with
        // the matrices' constant last column / w component are kept (matrix.h:58-65 forms all four per row)
        const NodeMotion& node = sc.animations[(size_t)o.camera_animation_id];
        float m[12], r[3][3];
        node.matrix3x4(time, m);
        node.rotation3x3(time, r);
        auto point = [&](const float3_& v) { return float3_{ v.x * m[0] + v.y * m[1] + v.z * m[2] + 1.0f * m[3], v.x * m[4] + v.y * m[5] + v.z * m[6] + 1.0f * m[7],
                                                             v.x * m[8] + v.y * m[9] + v.z * m[10] + 1.0f * m[11] }; };
        auto vector = [&](const float3_& v) { return float3_{ v.x * r[0][0] + v.y * r[0][1] + v.z * r[0][2] + 0.0f * 0.0f, v.x * r[1][0] + v.y * r[1][1] + v.z * r[1][2] + 0.0f * 0.0f,
                                                              v.x * r[2][0] + v.y * r[2][1] + v.z * r[2][2] + 0.0f * 0.0f }; };
        pos = point(cpos);
        dir = vector(cdir);
        up = vector(float3_{ 0, 1, 0 });
        right = normalize3(cross3(dir, up));
    } else {
        pos = cpos;
        dir = cdir;
        right = cross3(dir, float3_{ 0, 1, 0 }); // not normalised, sic (:1166)
        up = cross3(right, dir);
    }
    cam.pos[0] = pos.x; cam.pos[1] = pos.y; cam.pos[2] = pos.z;
    cam.dir[0] = dir.x; cam.dir[1] = dir.y; cam.dir[2] = dir.z;
    cam.up[0] = up.x; cam.up[1] = up.y; cam.up[2] = up.z;
    cam.right[0] = right.x; cam.right[1] = right.y; cam.right[2] = right.z;
}

} // namespace hjr

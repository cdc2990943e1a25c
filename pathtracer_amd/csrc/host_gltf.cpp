// glTF 2.0 / GLB scene reader: the part of the reference's model import that reaches the path
// tracer (Source/Runtime/Asset/ModelLoader.cpp:83-121 Load, :123-148 GenerateSubmesh,
// :150-287 GenerateMaterial), without assimp.
//
// What the reference's importer chain does and this reader reproduces:
//   * one sub-mesh per glTF mesh primitive, in mesh order then primitive order; node transforms and
//     the scene graph are ignored (the reference never applies them: ModelLoader.cpp:100-113);
//   * vertices are {position, normal, texcoord0}; the importer the reference uses flips the V
//     texture coordinate of glTF files (v -> 1 - v) and the reference loads its images flipped
//     vertically (PathTracingApplication.cpp:212), so both are kept;
//   * a primitive without NORMAL gets face normals (aiProcess_GenNormals: every vertex takes the
//     normal of the last face that references it); a primitive without TEXCOORD_0 is an error, as in
//     the reference (ModelLoader.cpp:389-403 throws);
//   * material attributes start from Material::MaterialAttributes' defaults (Material.h:43-59) and
//     are overwritten only by the properties the importer defines for the file: base colour,
//     metallic and roughness factors and the emissive factor always (glTF defaults 1,1,1 / 1 / 1 /
//     0,0,0), the KHR_materials_{emissive_strength, specular, sheen, clearcoat, ior, transmission}
//     values only when the extension is present; a primitive without a material gets the glTF
//     default material (white, metallic 1, roughness 1).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include "host.h"

namespace ptcore {
namespace {

// ---------------------------------------------------------------------------------- JSON
struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const char* key) const
    {
        if (type != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool isNumber() const { return type == Number; }
};

struct JsonParser {
    const char* p;
    const char* end;
    std::string err;

    void skip()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool fail(const std::string& what)
    {
        if (err.empty()) err = what;
        return false;
    }
    static void appendUtf8(std::string& s, unsigned cp)
    {
        if (cp < 0x80) {
            s.push_back((char)cp);
        } else if (cp < 0x800) {
            s.push_back((char)(0xC0 | (cp >> 6)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            s.push_back((char)(0xE0 | (cp >> 12)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool parseString(std::string& out)
    {
        if (p >= end || *p != '"') return fail("expected a string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("unterminated escape");
                switch (*p) {
                case 'n': out.push_back('\n'); break;
                case 't': out.push_back('\t'); break;
                case 'r': out.push_back('\r'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'u': {
                    if (end - p < 5) return fail("short \\u escape");
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; i++) {
                        const char c = p[i];
                        cp <<= 4;
                        if (c >= '0' && c <= '9') cp |= (unsigned)(c - '0');
                        else if (c >= 'a' && c <= 'f') cp |= (unsigned)(c - 'a' + 10);
                        else if (c >= 'A' && c <= 'F') cp |= (unsigned)(c - 'A' + 10);
                        else return fail("bad \\u escape");
                    }
                    appendUtf8(out, cp);
                    p += 4;
                    break;
                }
                default: out.push_back(*p); break;  // \" \\ \/
                }
                p++;
            } else {
                out.push_back(*p++);
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool parseValue(Json& v, int depth)
    {
        if (depth > 64) return fail("JSON nested too deeply");
        skip();
        if (p >= end) return fail("unexpected end of JSON");
        const char c = *p;
        if (c == '{') {
            v.type = Json::Object;
            p++;
            skip();
            if (p < end && *p == '}') {
                p++;
                return true;
            }
            for (;;) {
                skip();
                std::string key;
                if (!parseString(key)) return false;
                skip();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                v.obj.emplace_back(std::move(key), Json{});
                if (!parseValue(v.obj.back().second, depth + 1)) return false;
                skip();
                if (p < end && *p == ',') {
                    p++;
                    continue;
                }
                if (p < end && *p == '}') {
                    p++;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.type = Json::Array;
            p++;
            skip();
            if (p < end && *p == ']') {
                p++;
                return true;
            }
            for (;;) {
                v.arr.emplace_back();
                if (!parseValue(v.arr.back(), depth + 1)) return false;
                skip();
                if (p < end && *p == ',') {
                    p++;
                    continue;
                }
                if (p < end && *p == ']') {
                    p++;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.type = Json::String;
            return parseString(v.str);
        }
        if (end - p >= 4 && std::strncmp(p, "true", 4) == 0) {
            v.type = Json::Bool;
            v.b = true;
            p += 4;
            return true;
        }
        if (end - p >= 5 && std::strncmp(p, "false", 5) == 0) {
            v.type = Json::Bool;
            p += 5;
            return true;
        }
        if (end - p >= 4 && std::strncmp(p, "null", 4) == 0) {
            p += 4;
            return true;
        }
        if (c == '-' || (c >= '0' && c <= '9')) {
            std::string text;
            while (p < end && *p != '\0' && (std::strchr("+-.eE", *p) || (*p >= '0' && *p <= '9'))) text.push_back(*p++);
            char* stop = nullptr;
            v.num = std::strtod(text.c_str(), &stop);
            if (stop == text.c_str() || *stop != '\0') return fail("malformed number '" + text + "'");
            v.type = Json::Number;
            return true;
        }
        return fail(std::string("unexpected character '") + c + "' in JSON");
    }
};

// ------------------------------------------------------------------------------- helpers
bool readFile(const std::string& path, std::vector<uint8_t>& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (n < 0) return false;
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0) f.read(reinterpret_cast<char*>(out.data()), n);
    return (bool)f;
}

bool decodeBase64(const std::string& text, size_t from, std::vector<uint8_t>& out)
{
    unsigned acc = 0;
    int bits = 0;
    for (size_t i = from; i < text.size(); i++) {
        const char c = text[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r') continue;
        else return false;
        acc = (acc << 6) | (unsigned)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)((acc >> bits) & 0xFFu));
        }
    }
    return true;
}

std::string percentDecode(const std::string& uri)
{
    std::string out;
    for (size_t i = 0; i < uri.size(); i++) {
        if (uri[i] == '%' && i + 2 < uri.size() && std::isxdigit((unsigned char)uri[i + 1]) &&
            std::isxdigit((unsigned char)uri[i + 2])) {
            out.push_back((char)std::strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16));
            i += 2;
        } else {
            out.push_back(uri[i]);
        }
    }
    return out;
}

int64_t getInt(const Json& o, const char* key, int64_t fallback)
{
    const Json* v = o.find(key);
    return (v && v->isNumber()) ? (int64_t)v->num : fallback;
}

bool getFloat(const Json* o, const char* key, float& out)
{
    if (!o) return false;
    const Json* v = o->find(key);
    if (!v || !v->isNumber()) return false;
    out = (float)v->num;
    return true;
}

bool getVec(const Json* o, const char* key, float* out, int n)
{
    if (!o) return false;
    const Json* v = o->find(key);
    if (!v || v->type != Json::Array || (int)v->arr.size() < n) return false;
    for (int i = 0; i < n; i++) {
        if (!v->arr[(size_t)i].isNumber()) return false;
        out[i] = (float)v->arr[(size_t)i].num;
    }
    return true;
}

float length3(const float c[3]) { return std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]); }  // glm::length

}  // namespace

// --------------------------------------------------------------------------------- scene
std::string LoadGltf(const std::string& path, GltfScene& scene)
{
    scene = GltfScene{};
    std::vector<uint8_t> file;
    if (!readFile(path, file)) return "cannot read '" + path + "'";
    std::string folder = path;
    const size_t slash = folder.find_last_of("/\\");
    folder = slash == std::string::npos ? std::string() : folder.substr(0, slash + 1);

    // GLB container: 12-byte header, then a JSON chunk and an optional BIN chunk
    const uint8_t* jsonBegin = file.data();
    size_t jsonSize = file.size();
    std::vector<uint8_t> glbBin;
    bool haveGlbBin = false;
    if (file.size() >= 12 && std::memcmp(file.data(), "glTF", 4) == 0) {
        uint32_t version, total;
        std::memcpy(&version, file.data() + 4, 4);
        std::memcpy(&total, file.data() + 8, 4);
        if (version != 2) return "GLB container version " + std::to_string(version) + " is not supported";
        if (total > file.size()) return "GLB header claims more bytes than the file holds";
        size_t off = 12;
        bool haveJson = false;
        while (off + 8 <= total) {
            uint32_t len, type;
            std::memcpy(&len, file.data() + off, 4);
            std::memcpy(&type, file.data() + off + 4, 4);
            off += 8;
            if (off + len > total) return "GLB chunk runs past the end of the file";
            if (type == 0x4E4F534Au && !haveJson) {
                jsonBegin = file.data() + off;
                jsonSize = len;
                haveJson = true;
            } else if (type == 0x004E4942u && !haveGlbBin) {
                glbBin.assign(file.data() + off, file.data() + off + len);
                haveGlbBin = true;
            }
            off += (len + 3u) & ~3u;
        }
        if (!haveJson) return "GLB file has no JSON chunk";
    }

    Json doc;
    JsonParser parser{reinterpret_cast<const char*>(jsonBegin), reinterpret_cast<const char*>(jsonBegin) + jsonSize, {}};
    if (!parser.parseValue(doc, 0)) return "JSON: " + parser.err;
    if (doc.type != Json::Object) return "glTF document is not a JSON object";
    const Json* asset = doc.find("asset");
    const Json* ver = asset ? asset->find("version") : nullptr;
    if (!ver || ver->type != Json::String || ver->str.compare(0, 2, "2.") != 0) return "not a glTF 2.x asset";

    auto arrayOf = [&](const char* key) -> const std::vector<Json>& {
        static const std::vector<Json> empty;
        const Json* a = doc.find(key);
        return (a && a->type == Json::Array) ? a->arr : empty;
    };
    const auto& jBuffers = arrayOf("buffers");
    const auto& jViews = arrayOf("bufferViews");
    const auto& jAccessors = arrayOf("accessors");
    const auto& jMeshes = arrayOf("meshes");
    const auto& jMaterials = arrayOf("materials");
    const auto& jTextures = arrayOf("textures");
    const auto& jImages = arrayOf("images");

    // ---- buffers
    std::vector<std::vector<uint8_t>> buffers(jBuffers.size());
    for (size_t i = 0; i < jBuffers.size(); i++) {
        const Json* uri = jBuffers[i].find("uri");
        if (!uri) {
            if (i == 0 && haveGlbBin) {
                buffers[i] = std::move(glbBin);
            } else {
                return "buffer " + std::to_string(i) + " has no uri and there is no GLB binary chunk for it";
            }
        } else if (uri->str.compare(0, 5, "data:") == 0) {
            const size_t comma = uri->str.find(',');
            if (comma == std::string::npos || uri->str.find(";base64") == std::string::npos || !decodeBase64(uri->str, comma + 1, buffers[i]))
                return "buffer " + std::to_string(i) + ": unsupported data URI";
        } else if (!readFile(folder + percentDecode(uri->str), buffers[i])) {
            return "cannot read buffer file '" + folder + percentDecode(uri->str) + "'";
        }
        const int64_t declared = getInt(jBuffers[i], "byteLength", -1);
        if (declared < 0 || (uint64_t)declared > buffers[i].size())
            return "buffer " + std::to_string(i) + " holds " + std::to_string(buffers[i].size()) + " bytes, byteLength says " + std::to_string(declared);
    }

    // ---- accessor reader: element `i`, component `c` as double
    struct View {
        const uint8_t* base = nullptr;
        size_t stride = 0, count = 0;
        int components = 0, componentType = 0;
        bool normalized = false;
    };
    auto openAccessor = [&](int64_t index, const char* what, View& v) -> std::string {
        if (index < 0 || (size_t)index >= jAccessors.size()) return std::string(what) + ": accessor index out of range";
        const Json& a = jAccessors[(size_t)index];
        if (a.find("sparse")) return std::string(what) + ": sparse accessors are not supported";
        const int64_t viewIndex = getInt(a, "bufferView", -1);
        if (viewIndex < 0 || (size_t)viewIndex >= jViews.size()) return std::string(what) + ": accessor without a valid bufferView";
        const Json& bv = jViews[(size_t)viewIndex];
        const int64_t bufIndex = getInt(bv, "buffer", -1);
        if (bufIndex < 0 || (size_t)bufIndex >= buffers.size()) return std::string(what) + ": bufferView without a valid buffer";
        const Json* type = a.find("type");
        if (!type || type->type != Json::String) return std::string(what) + ": accessor without a type";
        static const std::map<std::string, int> kComponents = {{"SCALAR", 1}, {"VEC2", 2}, {"VEC3", 3}, {"VEC4", 4}};
        const auto it = kComponents.find(type->str);
        if (it == kComponents.end()) return std::string(what) + ": accessor type " + type->str + " is not supported";
        v.components = it->second;
        v.componentType = (int)getInt(a, "componentType", 0);
        size_t compBytes;
        switch (v.componentType) {
        case 5120: case 5121: compBytes = 1; break;
        case 5122: case 5123: compBytes = 2; break;
        case 5125: case 5126: compBytes = 4; break;
        default: return std::string(what) + ": unknown componentType " + std::to_string(v.componentType);
        }
        const Json* norm = a.find("normalized");
        v.normalized = norm && norm->type == Json::Bool && norm->b;
        // Every number below comes from the file: reject negative values before any of them is cast, and
        // phrase the range check so that no sum or product can wrap.
        const int64_t count = getInt(a, "count", 0), byteStride = getInt(bv, "byteStride", 0);
        const int64_t viewOffset = getInt(bv, "byteOffset", 0), viewLength = getInt(bv, "byteLength", 0);
        const int64_t accessorOffset = getInt(a, "byteOffset", 0);
        if (count < 0 || byteStride < 0 || viewOffset < 0 || viewLength < 0 || accessorOffset < 0)
            return std::string(what) + ": negative count, stride, offset or length";
        v.count = (size_t)count;
        const size_t elemBytes = compBytes * (size_t)v.components;
        v.stride = byteStride > 0 ? (size_t)byteStride : elemBytes;
        const std::vector<uint8_t>& buf = buffers[(size_t)bufIndex];
        // the view must lie inside the buffer, the accessor's elements inside the view
        if ((size_t)viewOffset > buf.size() || (size_t)viewLength > buf.size() - (size_t)viewOffset)
            return std::string(what) + ": bufferView outside its buffer";
        if ((size_t)accessorOffset > (size_t)viewLength) return std::string(what) + ": accessor reads past its buffer view";
        const size_t start = (size_t)viewOffset + (size_t)accessorOffset;
        const size_t room = (size_t)viewLength - (size_t)accessorOffset;  // bytes from `start` to the view's end
        if (v.count > 0 && (elemBytes > room || (v.count - 1) > (room - elemBytes) / v.stride))
            return std::string(what) + ": accessor reads past its buffer view";
        v.base = buf.data() + start;
        return std::string();
    };
    auto component = [](const View& v, size_t i, int c) -> double {
        const uint8_t* p = v.base + i * v.stride;
        switch (v.componentType) {
        case 5120: { int8_t x; std::memcpy(&x, p + c, 1); return v.normalized ? std::fmax(x / 127.0, -1.0) : x; }
        case 5121: { uint8_t x; std::memcpy(&x, p + c, 1); return v.normalized ? x / 255.0 : x; }
        case 5122: { int16_t x; std::memcpy(&x, p + 2 * c, 2); return v.normalized ? std::fmax(x / 32767.0, -1.0) : x; }
        case 5123: { uint16_t x; std::memcpy(&x, p + 2 * c, 2); return v.normalized ? x / 65535.0 : x; }
        case 5125: { uint32_t x; std::memcpy(&x, p + 4 * c, 4); return x; }
        default: { float x; std::memcpy(&x, p + 4 * c, 4); return x; }
        }
    };

    // ---- images
    scene.images.resize(jImages.size());
    for (size_t i = 0; i < jImages.size(); i++) {
        GltfImage& img = scene.images[i];
        const Json* uri = jImages[i].find("uri");
        if (uri && uri->type == Json::String) {
            if (uri->str.compare(0, 5, "data:") == 0) {
                const size_t comma = uri->str.find(',');
                if (comma == std::string::npos || !decodeBase64(uri->str, comma + 1, img.bytes))
                    return "image " + std::to_string(i) + ": unsupported data URI";
            } else {
                img.path = folder + percentDecode(uri->str);
            }
        } else {
            const int64_t viewIndex = getInt(jImages[i], "bufferView", -1);
            if (viewIndex < 0 || (size_t)viewIndex >= jViews.size()) return "image " + std::to_string(i) + " has neither uri nor bufferView";
            const Json& bv = jViews[(size_t)viewIndex];
            const int64_t bufIndex = getInt(bv, "buffer", -1);
            const int64_t offSigned = getInt(bv, "byteOffset", 0), lenSigned = getInt(bv, "byteLength", 0);
            if (bufIndex < 0 || (size_t)bufIndex >= buffers.size() || offSigned < 0 || lenSigned < 0 ||
                (size_t)offSigned > buffers[(size_t)bufIndex].size() ||
                (size_t)lenSigned > buffers[(size_t)bufIndex].size() - (size_t)offSigned)
                return "image " + std::to_string(i) + ": bufferView outside its buffer";
            const size_t off = (size_t)offSigned, len = (size_t)lenSigned;
            img.bytes.assign(buffers[(size_t)bufIndex].begin() + (std::ptrdiff_t)off, buffers[(size_t)bufIndex].begin() + (std::ptrdiff_t)(off + len));
        }
    }
    auto textureImage = [&](const Json* holder, const char* key) -> int32_t {
        const Json* info = holder ? holder->find(key) : nullptr;
        if (!info) return -1;
        const int64_t tex = getInt(*info, "index", -1);
        if (tex < 0 || (size_t)tex >= jTextures.size()) return -1;
        const int64_t src = getInt(jTextures[(size_t)tex], "source", -1);
        return (src >= 0 && (size_t)src < jImages.size()) ? (int32_t)src : -1;
    };

    // ---- materials (ModelLoader.cpp:198-284)
    auto makeMaterial = [&](const Json* m) {
        GltfMaterial out;  // MaterialAttributes defaults
        float base[4] = {1.0f, 1.0f, 1.0f, 1.0f}, emissive[3] = {0.0f, 0.0f, 0.0f};
        float metallic = 1.0f, roughness = 1.0f;
        const Json* pbr = m ? m->find("pbrMetallicRoughness") : nullptr;
        const Json* ext = m ? m->find("extensions") : nullptr;
        getVec(pbr, "baseColorFactor", base, 4);
        getFloat(pbr, "metallicFactor", metallic);
        getFloat(pbr, "roughnessFactor", roughness);
        getVec(m, "emissiveFactor", emissive, 3);
        std::memcpy(out.emission, emissive, 12);                                     // :241
        const Json* strength = ext ? ext->find("KHR_materials_emissive_strength") : nullptr;
        if (strength) {                                                              // :244
            float s = 1.0f;
            getFloat(strength, "emissiveStrength", s);
            for (float& e : out.emission) e *= s;
        }
        std::memcpy(out.albedo, base, 12);                                           // :247
        const Json* spec = ext ? ext->find("KHR_materials_specular") : nullptr;
        if (spec) {
            float factor = 1.0f, color[3] = {1.0f, 1.0f, 1.0f};
            getFloat(spec, "specularFactor", factor);
            getVec(spec, "specularColorFactor", color, 3);
            out.specular = factor;                                                   // :250
            out.specularTint = length3(color);                                       // :253
        }
        out.metallic = metallic;                                                     // :256
        out.roughness = roughness;                                                   // :259
        const Json* sheen = ext ? ext->find("KHR_materials_sheen") : nullptr;
        if (sheen) {
            float r = 0.0f, color[3] = {0.0f, 0.0f, 0.0f};
            getFloat(sheen, "sheenRoughnessFactor", r);
            getVec(sheen, "sheenColorFactor", color, 3);
            out.sheenRoughness = r;                                                  // :266
            out.sheenTint = length3(color);                                          // :269
        }
        const Json* coat = ext ? ext->find("KHR_materials_clearcoat") : nullptr;
        if (coat) {
            float f = 0.0f, r = 0.0f;
            getFloat(coat, "clearcoatFactor", f);
            getFloat(coat, "clearcoatRoughnessFactor", r);
            out.clearcoat = f;                                                       // :272
            out.clearcoatRoughness = r;                                              // :275
        }
        const Json* ior = ext ? ext->find("KHR_materials_ior") : nullptr;
        if (ior) {
            float v = 1.5f;
            getFloat(ior, "ior", v);
            out.refraction = v;                                                      // :278
        }
        const Json* trans = ext ? ext->find("KHR_materials_transmission") : nullptr;
        if (trans) {
            float v = 0.0f;
            getFloat(trans, "transmissionFactor", v);
            out.transmission = v;                                                    // :281
        }
        // texture slots in Material::MaterialTextureSlot order (:200-231).  The specular-colour slot
        // asks the importer for texture 0 of the specular type like the specular slot does (the
        // index argument is dropped at ModelLoader.cpp:320), so both slots get the same image.
        out.image[0] = textureImage(m, "emissiveTexture");
        out.image[1] = textureImage(pbr, "baseColorTexture");
        out.image[2] = textureImage(m, "normalTexture");
        int32_t specImage = textureImage(spec, "specularTexture");
        if (specImage < 0) specImage = textureImage(spec, "specularColorTexture");
        out.image[3] = specImage;
        out.image[4] = specImage;
        out.image[5] = textureImage(pbr, "metallicRoughnessTexture");
        out.image[6] = textureImage(sheen, "sheenRoughnessTexture");
        out.image[7] = textureImage(sheen, "sheenColorTexture");
        out.image[8] = textureImage(coat, "clearcoatTexture");
        out.image[9] = textureImage(coat, "clearcoatRoughnessTexture");
        out.image[10] = textureImage(trans, "transmissionTexture");
        return out;
    };

    // ---- sub-meshes
    for (size_t mi = 0; mi < jMeshes.size(); mi++) {
        const Json* prims = jMeshes[mi].find("primitives");
        if (!prims || prims->type != Json::Array) continue;
        for (size_t pi = 0; pi < prims->arr.size(); pi++) {
            const Json& prim = prims->arr[pi];
            const std::string where = "mesh " + std::to_string(mi) + " primitive " + std::to_string(pi);
            const int64_t mode = getInt(prim, "mode", 4);
            if (mode != 4) return where + ": only triangle lists (mode 4) are supported, mode is " + std::to_string(mode);
            const Json* attrs = prim.find("attributes");
            if (!attrs) return where + " has no attributes";
            const int64_t posIndex = getInt(*attrs, "POSITION", -1), norIndex = getInt(*attrs, "NORMAL", -1),
                          uvIndex = getInt(*attrs, "TEXCOORD_0", -1);
            if (posIndex < 0) return where + " has no POSITION";
            if (uvIndex < 0) return where + " has no TEXCOORD_0 (the reference's loader requires it)";
            View pos, nor, uv, idx;
            std::string err = openAccessor(posIndex, (where + " POSITION").c_str(), pos);
            if (err.empty()) err = openAccessor(uvIndex, (where + " TEXCOORD_0").c_str(), uv);
            if (err.empty() && norIndex >= 0) err = openAccessor(norIndex, (where + " NORMAL").c_str(), nor);
            if (!err.empty()) return err;
            if (pos.components != 3 || pos.componentType != 5126) return where + ": POSITION must be float VEC3";
            if (uv.components != 2) return where + ": TEXCOORD_0 must be VEC2";
            if (uv.count != pos.count || (norIndex >= 0 && (nor.count != pos.count || nor.components != 3)))
                return where + ": attribute counts differ";

            GltfSubmesh sm;
            sm.vertices.resize(pos.count);
            for (size_t i = 0; i < pos.count; i++) {
                RefVertex& v = sm.vertices[i];
                for (int c = 0; c < 3; c++) v.pos[c] = (float)component(pos, i, c);
                for (int c = 0; c < 3; c++) v.nor[c] = norIndex >= 0 ? (float)component(nor, i, c) : 0.0f;
                v.uv[0] = (float)component(uv, i, 0);
                v.uv[1] = 1.0f - (float)component(uv, i, 1);
            }
            const int64_t idxIndex = getInt(prim, "indices", -1);
            if (idxIndex >= 0) {
                err = openAccessor(idxIndex, (where + " indices").c_str(), idx);
                if (!err.empty()) return err;
                if (idx.components != 1 || (idx.componentType != 5121 && idx.componentType != 5123 && idx.componentType != 5125))
                    return where + ": indices must be unsigned SCALAR";
                sm.indices.resize(idx.count);
                for (size_t i = 0; i < idx.count; i++) {
                    const double x = component(idx, i, 0);
                    if (x < 0.0 || x >= (double)pos.count) return where + ": index " + std::to_string((uint64_t)x) + " is outside the vertex array";
                    sm.indices[i] = (uint32_t)x;
                }
            } else {
                sm.indices.resize(pos.count);
                for (size_t i = 0; i < pos.count; i++) sm.indices[i] = (uint32_t)i;
            }
            if (sm.indices.size() % 3 != 0) return where + ": index count is not a multiple of 3";
            if (norIndex < 0) {  // face normals, last face wins on shared vertices
                for (size_t f = 0; f + 2 < sm.indices.size(); f += 3) {
                    const float* a = sm.vertices[sm.indices[f]].pos;
                    const float* b = sm.vertices[sm.indices[f + 1]].pos;
                    const float* c = sm.vertices[sm.indices[f + 2]].pos;
                    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    const float len = length3(n);
                    if (len > 0.0f)
                        for (float& x : n) x /= len;
                    for (int k = 0; k < 3; k++) std::memcpy(sm.vertices[sm.indices[f + (size_t)k]].nor, n, 12);
                }
            }
            const int64_t matIndex = getInt(prim, "material", -1);
            if (matIndex >= (int64_t)jMaterials.size()) return where + ": material index out of range";
            sm.material = makeMaterial(matIndex >= 0 ? &jMaterials[(size_t)matIndex] : nullptr);
            scene.submeshes.push_back(std::move(sm));
        }
    }
    if (scene.submeshes.empty()) return "'" + path + "' holds no triangle meshes";
    return std::string();
}

}  // namespace ptcore

// pixie_amd/csrc/unet_exec.hip -- one U-Net as a C-ABI handle: pixie_unet_{create,set_param,workspace_bytes,forward,destroy}.
//
// Replaces, for one network, what the reference does with an nn.Module: construction (MyUNetModel.__init__,
// WG/models/module/diffusion_network.py:712-873; FeatureProjector :534-589; the wrappers trainer/training_discrete.py:50-88,
// trainer/training_continuous_mse.py:48-89), load_state_dict (keys in registration order) and forward (:875-935).  A
// forward pass is a fixed sequence of ~400 launches of the operators of section (A); this file owns that sequence, so a
// caller in any language pays ONE foreign call per scene instead of ~400 plus ~600 temporary allocations, and the sequence
// is capturable into a HIP graph (it allocates nothing and never synchronises after the first call).
//
// Memory: every temporary of a pass -- activations, statistics, split-K scratch -- lives in ONE caller-provided workspace
// (pixie_unet_workspace_bytes); a first-fit free list places them, tensors return their range when the last consumer has
// been launched (everything is ordered on one stream, so reuse is safe).  The placement is a function of the shapes only:
// the dry run that sizes the workspace and the real pass take the same decisions.  The handle owns only the re-packed
// weights.  The operators are called through the same extern "C" entry points a foreign caller would use; the launch order
// equals pixie_amd/unet.py: UNetRunner.forward, so both executors produce bit-identical tensors (tests/test_unet_hip.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/pixie_hip.h"
#include "common.h"

namespace pixie {
// internal entry points of unet_ops.hip (the launches of pixie_channel_stats / pixie_norm_finalize without their set-up work)
int channel_stats_prezeroed(const float* d_x, int channels, int64_t spatial, double* d_sums, uint32_t* d_amax, hipStream_t st);
int norm_finalize_cat(const double* d_sums0, int c0, const double* d_sums1, int c1, int64_t spatial, int mode, int groups, double eps,
                      const float* d_weight, const float* d_bias, float* d_a, float* d_b, hipStream_t st);
namespace {

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SILU = 2 };
enum Kind { CONV_IN, RES, DOWN, ATTN, UP };

struct Blk { Kind kind; std::string prefix; int cin, cout, sp; };
struct Plan {
    std::vector<std::vector<Blk>> in_blocks, out_blocks;
    std::vector<Blk> middle;
    int out_sp = 0;
};

struct Err : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] void fail(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw Err(buf);
}
void ok(int rc, const char* what) { if (rc != 0) fail("%s: %s", what, pixie_last_error()); }

bool has(const std::vector<int>& v, int x) { for (int e : v) if (e == x) return true; return false; }

struct Cfg {
    int feature_channels, cond_dim, model_channels, num_res_blocks;
    std::vector<int> mult, attn;
    int grid_size, out_channels, precision;
    bool has_projector() const { return feature_channels != cond_dim; }           // training_discrete.py:63-69
    int projector_hidden() const { return feature_channels > cond_dim ? 128 : 0; }  // 0: the light (1x1 + GroupNorm) projector
};

// diffusion_network.py:760-873, the constructor loop as data
Plan build_plan(const Cfg& c) {
    Plan p;
    const int mc = c.model_channels, nrb = c.num_res_blocks;
    auto name = [](const char* part, int idx, int sub) { return std::string("unet.") + part + "." + std::to_string(idx) + "." + std::to_string(sub); };
    p.in_blocks.push_back({Blk{CONV_IN, "unet.input_blocks.0.0", c.cond_dim, mc, c.grid_size}});
    std::vector<int> chans{mc}, sizes{c.grid_size};
    int ch = mc, ds = 1, sp = c.grid_size;
    for (size_t level = 0; level < c.mult.size(); ++level) {
        const int m = c.mult[level];
        for (int r = 0; r < nrb; ++r) {
            const int idx = (int)p.in_blocks.size();
            std::vector<Blk> seq{Blk{RES, name("input_blocks", idx, 0), ch, m * mc, sp}};
            ch = m * mc;
            if (has(c.attn, ds)) seq.push_back(Blk{ATTN, name("input_blocks", idx, 1), ch, ch, sp});
            p.in_blocks.push_back(seq);
            chans.push_back(ch);
        }
        if (level + 1 != c.mult.size()) {
            const int idx = (int)p.in_blocks.size();
            p.in_blocks.push_back({Blk{DOWN, name("input_blocks", idx, 0), ch, ch, sp}});
            chans.push_back(ch);
            sizes.push_back(sp);
            ds *= 2;
            sp = (sp + 1) / 2;
        }
    }
    p.middle = {Blk{RES, "unet.middle_block.0", ch, ch, sp}, Blk{ATTN, "unet.middle_block.1", ch, ch, sp}, Blk{RES, "unet.middle_block.2", ch, ch, sp}};
    for (int level = (int)c.mult.size() - 1; level >= 0; --level) {
        const int m = c.mult[level];
        for (int i = 0; i <= nrb; ++i) {
            const int idx = (int)p.out_blocks.size();
            const int ich = chans.back(); chans.pop_back();
            std::vector<Blk> seq{Blk{RES, name("output_blocks", idx, 0), ch + ich, mc * m, sp}};
            ch = mc * m;
            if (has(c.attn, ds)) seq.push_back(Blk{ATTN, name("output_blocks", idx, (int)seq.size()), ch, ch, sp});
            if (level && i == nrb) {
                seq.push_back(Blk{UP, name("output_blocks", idx, (int)seq.size()), ch, ch, sp});
                ds /= 2;
                sp = sizes.back(); sizes.pop_back();
            }
            p.out_blocks.push_back(seq);
        }
    }
    p.out_sp = sp;
    return p;
}

struct Param {
    std::string key;
    std::vector<int64_t> shape;
    int64_t numel = 1;
    bool norm = false;
    const float* d = nullptr;     // caller-owned device memory
    uint64_t version = 0;         // bumped by every pixie_unet_set_param
};

// state_dict keys and shapes in the reference's registration order
std::vector<Param> param_table(const Cfg& c, const Plan& plan) {
    std::vector<Param> out;
    auto add = [&](const std::string& key, std::vector<int64_t> shape, bool norm) {
        Param p; p.key = key; p.shape = shape; p.norm = norm;
        for (int64_t s : shape) p.numel *= s;
        out.push_back(p);
    };
    auto conv = [&](const std::string& prefix, int cout, int cin, int k, int dims = 3) {
        std::vector<int64_t> s{cout, cin};
        for (int i = 0; i < dims; ++i) s.push_back(k);
        add(prefix + ".weight", s, false);
        add(prefix + ".bias", {cout}, false);
    };
    auto norm = [&](const std::string& prefix, std::vector<int64_t> shape) {
        add(prefix + ".weight", shape, true);
        add(prefix + ".bias", shape, true);
    };
    if (c.has_projector()) {   // diffusion_network.py:556-585
        const int hid = c.projector_hidden();
        if (!hid) {
            conv("projector.net.0", c.cond_dim, c.feature_channels, 1);
            norm("projector.net.1", {c.cond_dim});
        } else {
            conv("projector.net.0", hid, c.feature_channels, 1);
            norm("projector.net.1", {hid});
            conv("projector.net.3", hid, hid, 3);
            norm("projector.net.4", {hid});
            conv("projector.net.6", c.cond_dim, hid, 1);
            norm("projector.net.7", {c.cond_dim});
        }
    }
    auto emit = [&](const Blk& b) {
        const std::vector<int64_t> vol{b.sp, b.sp, b.sp};
        switch (b.kind) {
            case CONV_IN: conv(b.prefix, b.cout, b.cin, 3); break;
            case RES:     // :673-694
                norm(b.prefix + ".in_layers.0", vol);
                conv(b.prefix + ".in_layers.2", b.cout, b.cin, 3);
                norm(b.prefix + ".out_layers.0", vol);
                conv(b.prefix + ".out_layers.3", b.cout, b.cout, 3);
                if (b.cin != b.cout) conv(b.prefix + ".skip_connection", b.cout, b.cin, 1);
                break;
            case DOWN: conv(b.prefix + ".op", b.cout, b.cin, 3); break;
            case UP:   conv(b.prefix + ".conv", b.cout, b.cin, 3); break;
            case ATTN:    // :199-208
                norm(b.prefix + ".norm", {b.cin});
                conv(b.prefix + ".qkv", 3 * b.cin, b.cin, 1, 1);
                conv(b.prefix + ".proj_out", b.cin, b.cin, 1, 1);
                break;
        }
    };
    for (auto& seq : plan.in_blocks) for (auto& b : seq) emit(b);
    for (auto& b : plan.middle) emit(b);
    for (auto& seq : plan.out_blocks) for (auto& b : seq) emit(b);
    norm("unet.out.0", {plan.out_sp, plan.out_sp, plan.out_sp});
    conv("unet.out.2", c.out_channels, c.model_channels, 3);
    return out;
}

// First-fit free list over [0, capacity) in units of bytes; 256-byte granules.  `base == nullptr` = dry run (offsets only).
struct Arena {
    static constexpr int64_t kAlign = 256;
    char* base = nullptr;
    int64_t capacity = 0, peak = 0;
    std::map<int64_t, int64_t> free_;    // offset -> size, disjoint, coalesced
    std::map<int64_t, int64_t> used_;    // offset -> size

    void reset(char* b, int64_t cap) {
        base = b; capacity = cap; peak = 0;
        free_.clear(); used_.clear();
        free_[0] = cap;
    }
    int64_t alloc(int64_t bytes) {
        bytes = (bytes + kAlign - 1) / kAlign * kAlign;
        if (bytes == 0) bytes = kAlign;
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second < bytes) continue;
            const int64_t off = it->first, rest = it->second - bytes;
            free_.erase(it);
            if (rest > 0) free_[off + bytes] = rest;
            used_[off] = bytes;
            if (off + bytes > peak) peak = off + bytes;
            return off;
        }
        fail("pixie_unet_forward: the workspace is too small (%lld bytes; ask pixie_unet_workspace_bytes for this grid size)", (long long)capacity);
    }
    void release(int64_t off) {
        auto u = used_.find(off);
        if (u == used_.end()) return;
        int64_t size = u->second;
        used_.erase(u);
        auto next = free_.lower_bound(off);
        if (next != free_.end() && off + size == next->first) { size += next->second; next = free_.erase(next); }
        if (next != free_.begin()) {
            auto prev = std::prev(next);
            if (prev->first + prev->second == off) { prev->second += size; return; }
        }
        free_[off] = size;
    }
    template <class T> T* ptr(int64_t off) const { return reinterpret_cast<T*>(base + off); }   // dry run: small fake addresses, never dereferenced
};

struct Exec;
struct Tens {
    Exec* ex = nullptr;
    float* p = nullptr;
    int64_t off = -1;             // arena offset, -1: caller-owned
    int c = 0, d = 0, h = 0, w = 0;
    int64_t sums_off = -1;        // arena range of `sums` when a conv epilogue produced them (-1: pre-zeroed region, or none yet)
    bool has_sums = false;        // double[2c] channel sums of this tensor exist (somebody needed them)
    double* sums = nullptr;
    uint32_t* slot = nullptr;     // |x|max as float bits
    int64_t spatial() const { return (int64_t)d * h * w; }
    ~Tens();
};
using TP = std::shared_ptr<Tens>;

struct Packed { void* d = nullptr; uint64_t version = ~0ull; };
struct Sized { int64_t bytes; int slots; int64_t zsum_doubles; };   // workspace bytes, |x|max words, pre-zeroed statistics doubles
struct Bound { float wmax = 0, bmax = 0; uint64_t wver = ~0ull, bver = ~0ull; };

}  // namespace
}  // namespace pixie

using namespace pixie;

struct pixie_unet {
    Cfg cfg;
    Plan plan;
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    std::unordered_map<std::string, Packed> packed32, packed16;
    std::unordered_map<std::string, Bound> bounds;
    uint32_t* d_bound_slots = nullptr;     // 2 words per normalisation layer, device
    std::vector<uint32_t> h_bound_slots;
    std::map<std::tuple<int, int, int, bool>, Sized> sized;   // (d, h, w, starts behind projector.net[0]) -> what one pass needs
    bool fuse_stats = true, split_k = true, fold_skip = true;
    // pixie_unet_set_option("graph", 1): forward() replays a captured HIP graph when called again with the same pointers
    struct Replay { const float* feat; const float* proj0; float* out; void* ws; int d; uint64_t epoch; hipGraph_t graph; hipGraphExec_t exec; };
    bool use_graph = false;
    uint64_t epoch = 0;                     // bumped by every set_param: captured kernel arguments (bounds, weights) are stale after it
    std::vector<Replay> replays;            // most recently used last; at most kMaxReplays
    static constexpr size_t kMaxReplays = 4;
    void drop_replay(size_t i) {
        (void)hipGraphExecDestroy(replays[i].exec);
        (void)hipGraphDestroy(replays[i].graph);
        replays.erase(replays.begin() + (long)i);
    }

    const Param& param(const std::string& key) const {
        auto it = index.find(key);
        if (it == index.end()) fail("pixie_unet: no parameter named '%s'", key.c_str());
        return params[it->second];
    }
};

namespace pixie {
namespace {

struct Exec {
    pixie_unet* net = nullptr;
    Arena arena;
    bool dry = false;
    void* stream = nullptr;
    // head of the workspace, zeroed by ONE memset per pass: |x|max words, then the double[2c] targets of the statistics passes
    uint32_t* slots = nullptr;
    int slot_next = 0, slot_cap = 0;
    double* zsums = nullptr;
    int64_t zsum_next = 0, zsum_cap = 0;

    TP make(int c, int d, int h, int w) {
        auto t = std::make_shared<Tens>();
        t->ex = this; t->c = c; t->d = d; t->h = h; t->w = w;
        t->off = arena.alloc((int64_t)c * d * h * w * sizeof(float));
        t->p = arena.ptr<float>(t->off);
        return t;
    }
    TP external(const float* p, int c, int d, int h, int w) {
        auto t = std::make_shared<Tens>();
        t->ex = this; t->p = const_cast<float*>(p); t->c = c; t->d = d; t->h = h; t->w = w;
        return t;
    }
    uint32_t* new_slot() {
        if (!dry && slot_next >= slot_cap) fail("pixie_unet_forward: out of |x|max slots (internal sizing error)");
        return dry ? (slot_next++, nullptr) : slots + slot_next++;
    }
    struct Scratch {   // a temporary that must outlive the launches that use it only in stream order: freed at once
        Exec* ex; int64_t off;
        Scratch(Exec* e, int64_t bytes) : ex(e), off(e->arena.alloc(bytes)) {}
        ~Scratch() { ex->arena.release(off); }
        template <class T> T* as() const { return ex->arena.ptr<T>(off); }
    };

    // ---- parameters ----
    const float* P(const std::string& key) {
        const Param& p = net->param(key);
        if (!dry && !p.d) fail("pixie_unet_forward: parameter '%s' was never set", key.c_str());
        return p.d;
    }
    const void* w16(const std::string& key) {     // f16 hi/lo packing, once per parameter version
        const Param& p = net->param(key + ".weight");
        if (dry) return nullptr;
        Packed& e = net->packed16[key];
        if (e.version != p.version) {
            const int cout = (int)p.shape[0], cin = (int)p.shape[1], k = (int)p.shape[2];
            const int64_t nb = pixie_conv_packed16_bytes(cout, cin, k);
            if (nb <= 0) fail("pixie_unet: '%s' cannot take the f16x3 packing (c_in %d)", key.c_str(), cin);
            if (!e.d && hipMalloc(&e.d, (size_t)nb) != hipSuccess) fail("pixie_unet: hipMalloc of %lld bytes failed", (long long)nb);
            if (!p.d) fail("pixie_unet_forward: parameter '%s.weight' was never set", key.c_str());
            ok(pixie_conv_pack_weights_f16x2(p.d, e.d, cout, cin, k, stream), "pixie_conv_pack_weights_f16x2");
            e.version = p.version;
        }
        return e.d;
    }
    const float* w32(const std::string& key) {    // [tap][c_in][c_out padded] fp32 packing
        const Param& p = net->param(key + ".weight");
        if (dry) return nullptr;
        Packed& e = net->packed32[key];
        if (e.version != p.version) {
            const int cout = (int)p.shape[0], cin = (int)p.shape[1], k = (int)p.shape[2];
            int64_t taps = 1;
            for (size_t i = 2; i < p.shape.size(); ++i) taps *= p.shape[i];
            const int64_t nb = taps * cin * pixie_conv_cout_padded(cout) * (int64_t)sizeof(float);
            if (!e.d && hipMalloc(&e.d, (size_t)nb) != hipSuccess) fail("pixie_unet: hipMalloc of %lld bytes failed", (long long)nb);
            if (!p.d) fail("pixie_unet_forward: parameter '%s.weight' was never set", key.c_str());
            ok(pixie_conv_pack_weights(p.d, static_cast<float*>(e.d), cout, cin, k, stream), "pixie_conv_pack_weights");
            e.version = p.version;
        }
        return static_cast<const float*>(e.d);
    }
    // Bound on |normalised * weight + bias| when `count` elements share the statistics: a standardised sample of n values
    // cannot exceed sqrt(n - 1) in magnitude; LeakyReLU / SiLU do not increase magnitudes.  (unet.py: _norm_bound)
    float norm_bound(const std::string& key, int64_t count) {
        if (dry) return 1.0f;
        const Bound& b = net->bounds[key];
        return (float)(std::sqrt((double)count) * (double)b.wmax + (double)b.bmax + 1e-30);
    }

    // ---- statistics ----
    void stats(const TP& t) {                  // channel sums + |x|max by a pass over the tensor (only where no conv epilogue produced them)
        if (t->has_sums) return;
        t->has_sums = true;
        if (!dry && zsum_next + 2 * t->c > zsum_cap) fail("pixie_unet_forward: out of statistics space (internal sizing error)");
        t->sums = dry ? nullptr : zsums + zsum_next;      // the kernel adds into zeroed memory (fp64 atomics)
        zsum_next += 2 * t->c;
        t->slot = new_slot();
        if (!dry) ok(channel_stats_prezeroed(t->p, t->c, t->spatial(), t->sums, t->slot, as_stream(stream)), "pixie_channel_stats");
    }
    struct AB { std::shared_ptr<Scratch> mem; float* a; float* b; };
    AB norm_finalize(const double* sums, int c, int64_t spatial, int mode, int groups, const float* w, const float* b) {
        AB r;
        const int cpad = (c + 63) / 64 * 64;
        r.mem = std::make_shared<Scratch>(this, (int64_t)(cpad + c) * sizeof(float));
        r.a = r.mem->as<float>();
        r.b = r.a + cpad;
        if (!dry) ok(pixie_norm_finalize(sums, c, spatial, mode, groups, 1e-5, w, b, r.a, r.b, stream), "pixie_norm_finalize");
        return r;
    }

    bool f16_ok(const std::vector<TP>& parts, int stride) const {    // unet.py: HipOps.f16x3_ok
        int cin = 0;
        for (auto& t : parts) cin += t->c;
        return net->cfg.precision == 0 && (stride == 1 || stride == 2) && cin % 16 == 0 && parts[0]->c % 8 == 0;
    }
    bool skip_foldable(const TP& x, int cout, int ksize, const std::vector<TP>& sp) const {   // unet.py: HipOps.skip_foldable
        pixie_conv_desc d;
        std::memset(&d, 0, sizeof d);
        d.c0 = x->c; d.in_d = x->d; d.in_h = x->h; d.in_w = x->w;
        d.stride = 1; d.ksize = ksize; d.c_out = cout;
        d.d_w16 = reinterpret_cast<const void*>(0x100);
        d.d_workspace = net->split_k ? reinterpret_cast<void*>(0x100) : nullptr;
        d.skip_c0 = sp[0]->c; d.skip_c1 = sp.size() > 1 ? sp[1]->c : 0;
        return pixie_conv_skip_foldable(&d) != 0;
    }

    // ---- one convolution launch (unet.py: UNetRunner._conv + HipOps.conv) ----
    struct ConvOpt {
        int stride = 1; bool upsample = false;
        const AB* pro = nullptr; std::string affine_store;     // affine_store: key of the spatial LayerNorm whose gamma/beta the prologue applies
        int act = ACT_NONE; TP residual; float bound = 0.0f;
        int out_d = 0, out_h = 0, out_w = 0;
        float* out_ptr = nullptr;                               // write the result here (caller-owned) instead of into the arena
        const std::vector<TP>* skip_parts = nullptr; std::string skip_key;   // folded 1x1x1 skip convolution over these raw tensors
    };
    TP conv(const std::vector<TP>& parts, const std::string& wkey, int cout, int ksize, const ConvOpt& o) {
        const TP& x0 = parts[0];
        const Tens* x1 = parts.size() > 1 ? parts[1].get() : nullptr;
        const int up = o.upsample ? 2 : 1, pad = ksize == 3 ? 1 : 0;
        int od = (x0->d * up + 2 * pad - ksize) / o.stride + 1;
        int oh = (x0->h * up + 2 * pad - ksize) / o.stride + 1;
        int ow = (x0->w * up + 2 * pad - ksize) / o.stride + 1;
        pixie_conv_desc desc;
        std::memset(&desc, 0, sizeof desc);
        if (o.out_d) {   // odd-grid crop (diffusion_network.py:925-930): the cropped voxels are never computed
            od = std::min(od, o.out_d); oh = std::min(oh, o.out_h); ow = std::min(ow, o.out_w);
            desc.out_d = od; desc.out_h = oh; desc.out_w = ow;
        }
        const bool f16 = f16_ok(parts, o.stride);
        const bool raw = !o.pro && o.affine_store.empty();
        if (f16 && raw) for (auto& t : parts) stats(t);       // the input scale comes from the tensors' device-side |x|max

        TP out = o.out_ptr ? external(o.out_ptr, cout, od, oh, ow) : make(cout, od, oh, ow);
        desc.d_in0 = x0->p; desc.c0 = x0->c;
        desc.d_in1 = x1 ? x1->p : nullptr; desc.c1 = x1 ? x1->c : 0;
        desc.in_d = x0->d; desc.in_h = x0->h; desc.in_w = x0->w;
        desc.upsample = o.upsample ? 1 : 0;
        desc.stride = o.stride;
        desc.ksize = ksize;
        if (o.pro) { desc.d_pro_a = o.pro->a; desc.d_pro_b = o.pro->b; }
        if (!o.affine_store.empty()) { desc.d_gamma = P(o.affine_store + ".weight"); desc.d_beta = P(o.affine_store + ".bias"); }
        desc.act = o.act;
        desc.d_bias = P(wkey + ".bias");
        desc.c_out = cout;
        desc.d_residual = o.residual ? o.residual->p : nullptr;
        desc.d_out = out->p;
        if (!f16) {
            desc.d_w = w32(wkey);
            if (!dry) ok(pixie_conv3d_forward(&desc, stream), "pixie_conv3d_forward");
            return out;
        }
        desc.d_w16 = w16(wkey);
        if (o.skip_parts) {
            const std::vector<TP>& sp = *o.skip_parts;
            desc.d_skip_in0 = sp[0]->p; desc.skip_c0 = sp[0]->c;
            desc.d_skip_in1 = sp.size() > 1 ? sp[1]->p : nullptr; desc.skip_c1 = sp.size() > 1 ? sp[1]->c : 0;
            desc.d_skip_w16 = w16(o.skip_key);
            desc.d_skip_bias = P(o.skip_key + ".bias");
            desc.d_skip_amax0 = sp[0]->slot; desc.d_skip_amax1 = sp.size() > 1 ? sp[1]->slot : nullptr;
            if (dry) desc.d_skip_w16 = reinterpret_cast<const void*>(0x100);
        }
        if (raw) {
            desc.d_in_amax0 = parts[0]->slot;
            desc.d_in_amax1 = x1 ? parts[1]->slot : nullptr;
        } else {
            desc.in_bound = o.bound;
        }
        // sizes of the optional buffers depend on the shape fields only; the dry run needs non-null placeholders for them
        std::unique_ptr<Scratch> workspace, tile_stats;
        if (dry) { desc.d_w16 = reinterpret_cast<const void*>(0x100); }
        if (net->split_k) {
            const int64_t wsb = pixie_conv_workspace_bytes(&desc);
            if (wsb > 0) { workspace.reset(new Scratch(this, wsb)); desc.d_workspace = workspace->as<void>(); if (dry) desc.d_workspace = reinterpret_cast<void*>(0x100); }
        }
        bool have_stats = false;
        if (net->fuse_stats) {
            // the output's channel sums and |x|max come out of the conv epilogue: no separate pass over the tensor
            uint32_t* slot = new_slot();
            const int64_t nfl = pixie_conv_stats_floats(&desc);
            if (nfl > 0) {
                tile_stats.reset(new Scratch(this, nfl * (int64_t)sizeof(float)));
                desc.d_out_stats = tile_stats->as<float>();
                desc.d_out_amax = slot;
                have_stats = true;
                out->slot = slot;
            }
        }
        if (!dry) ok(pixie_conv3d_forward(&desc, stream), "pixie_conv3d_forward");
        if (have_stats) {
            out->sums_off = arena.alloc((int64_t)cout * 2 * sizeof(double));
            out->sums = arena.ptr<double>(out->sums_off);
            out->has_sums = true;
            if (!dry) ok(pixie_stats_finalize(desc.d_out_stats, &desc, out->sums, stream), "pixie_stats_finalize");
        }
        return out;
    }

    // ---- blocks ----
    AB norm_finalize_parts(const std::vector<TP>& parts, int64_t spatial) {   // LayerNorm statistics of th.cat(parts): read where they lie
        for (auto& t : parts) stats(t);
        if (parts.size() == 1) return norm_finalize(parts[0]->sums, parts[0]->c, spatial, 0, 1, nullptr, nullptr);
        const int c0 = parts[0]->c, c1 = parts[1]->c, c = c0 + c1;
        AB r;
        const int cpad = (c + 63) / 64 * 64;
        r.mem = std::make_shared<Scratch>(this, (int64_t)(cpad + c) * sizeof(float));
        r.a = r.mem->as<float>();
        r.b = r.a + cpad;
        if (!dry) ok(norm_finalize_cat(parts[0]->sums, c0, parts[1]->sums, c1, spatial, 0, 1, 1e-5, nullptr, nullptr, r.a, r.b, as_stream(stream)),
                     "pixie_norm_finalize");
        return r;
    }
    TP res(const Blk& b, const std::vector<TP>& parts) {   // MyResBlock.forward, diffusion_network.py:696-705
        const std::string& p = b.prefix;
        const int64_t spatial = parts[0]->spatial();
        AB pro = norm_finalize_parts(parts, spatial);
        ConvOpt o1; o1.pro = &pro; o1.affine_store = p + ".in_layers.0"; o1.act = ACT_LEAKY; o1.bound = norm_bound(p + ".in_layers.0", spatial);
        TP h = conv(parts, p + ".in_layers.2", b.cout, 3, o1);
        stats(h);
        AB pro2 = norm_finalize(h->sums, b.cout, spatial, 0, 1, nullptr, nullptr);
        TP skip = parts[0];
        ConvOpt o2; o2.pro = &pro2; o2.affine_store = p + ".out_layers.0"; o2.act = ACT_LEAKY; o2.bound = norm_bound(p + ".out_layers.0", spatial);
        if (b.cin != b.cout) {
            if (net->fold_skip && f16_ok({h}, 1) && f16_ok(parts, 1) && skip_foldable(h, b.cout, 3, parts)) {
                // out = conv(h) + skip_connection(x) in ONE launch: the 1x1x1 convolution rides in the accumulators of the second
                // 3^3 convolution, the skip tensor never exists (conv3d_f16x3.hip, "folded skip")
                skip.reset();
                for (auto& t : parts) stats(t);
                o2.skip_parts = &parts; o2.skip_key = p + ".skip_connection";
            } else {
                skip = conv(parts, p + ".skip_connection", b.cout, 1, ConvOpt{});
            }
        }
        o2.residual = skip;
        return conv({h}, p + ".out_layers.3", b.cout, 3, o2);
    }
    TP attn(const Blk& b, const TP& x) {   // AttentionBlock._forward, diffusion_network.py:213-221
        const std::string& p = b.prefix;
        const int c = x->c;
        const int64_t spatial = x->spatial();
        stats(x);
        AB pro = norm_finalize(x->sums, c, spatial, 1, 32, P(p + ".norm.weight"), P(p + ".norm.bias"));
        ConvOpt o; o.pro = &pro; o.bound = norm_bound(p + ".norm", spatial * (c / 32));
        TP qkv = conv({x}, p + ".qkv", 3 * c, 1, o);
        TP att = make(c, x->d, x->h, x->w);
        if (!dry) ok(pixie_attention_forward(qkv->p, att->p, c, (int)spatial, stream), "pixie_attention_forward");
        qkv.reset();
        ConvOpt o2; o2.residual = x;
        return conv({att}, p + ".proj_out", c, 1, o2);
    }
    TP block(const Blk& b, const std::vector<TP>& parts, const Tens* up_to) {
        if (b.kind == RES) return res(b, parts);
        const TP& x = parts[0];
        if (b.kind == ATTN) return attn(b, x);
        if (b.kind == DOWN) { ConvOpt o; o.stride = 2; return conv({x}, b.prefix + ".op", b.cout, 3, o); }
        if (b.kind == UP) {
            // `up_to`: the skip tensor this output will be concatenated with; on odd grids it is one voxel smaller than
            // 2 * sp per axis and the reference crops h[..., :-1] (diffusion_network.py:925-930)
            ConvOpt o; o.upsample = true;
            if (up_to) { o.out_d = up_to->d; o.out_h = up_to->h; o.out_w = up_to->w; }
            return conv({x}, b.prefix + ".conv", b.cout, 3, o);
        }
        fail("pixie_unet: unexpected block kind");
    }

    TP forward(const float* d_feat, const float* d_proj0, int D, int H, int W, float* d_out) {
        const Cfg& c = net->cfg;
        const int64_t spatial = (int64_t)D * H * W;
        TP x;
        AB pro_in; const AB* pro_in_p = nullptr;
        int act_in = ACT_NONE; float bound_in = 0.0f;
        if (c.has_projector()) {   // FeatureProjector.net, diffusion_network.py:556-585
            const std::string q = "projector.net.";
            const int hid = c.projector_hidden();
            if (!hid) {
                const int g = std::max(c.cond_dim / 2, 1);
                x = conv({external(d_feat, c.feature_channels, D, H, W)}, q + "0", c.cond_dim, 1, ConvOpt{});
                stats(x);
                pro_in = norm_finalize(x->sums, c.cond_dim, spatial, 1, g, P(q + "1.weight"), P(q + "1.bias"));
                act_in = ACT_SILU;
                bound_in = norm_bound(q + "1", spatial * (c.cond_dim / g));
            } else {
                if (d_proj0) x = external(d_proj0, hid, D, H, W);
                else x = conv({external(d_feat, c.feature_channels, D, H, W)}, q + "0", hid, 1, ConvOpt{});
                stats(x);
                AB pro = norm_finalize(x->sums, hid, spatial, 1, 32, P(q + "1.weight"), P(q + "1.bias"));
                ConvOpt o; o.pro = &pro; o.act = ACT_SILU; o.bound = norm_bound(q + "1", spatial * (hid / 32));
                x = conv({x}, q + "3", hid, 3, o);
                stats(x);
                AB pro2 = norm_finalize(x->sums, hid, spatial, 1, 32, P(q + "4.weight"), P(q + "4.bias"));
                ConvOpt o2; o2.pro = &pro2; o2.act = ACT_SILU; o2.bound = norm_bound(q + "4", spatial * (hid / 32));
                x = conv({x}, q + "6", c.cond_dim, 1, o2);
                stats(x);
                pro_in = norm_finalize(x->sums, c.cond_dim, spatial, 1, 32, P(q + "7.weight"), P(q + "7.bias"));
                bound_in = norm_bound(q + "7", spatial * std::max(c.cond_dim / 32, 1));
            }
            pro_in_p = &pro_in;
        } else {
            x = external(d_feat, c.feature_channels, D, H, W);
        }
        const Plan& plan = net->plan;
        std::vector<TP> hs;
        const Blk& first = plan.in_blocks[0][0];
        ConvOpt of; of.pro = pro_in_p; of.act = act_in; of.bound = bound_in;
        TP h = conv({x}, first.prefix, first.cout, 3, of);
        x.reset();
        pro_in = AB{};
        hs.push_back(h);
        for (size_t i = 1; i < plan.in_blocks.size(); ++i) {
            for (const Blk& b : plan.in_blocks[i]) h = block(b, {h}, nullptr);
            hs.push_back(h);
        }
        for (const Blk& b : plan.middle) h = block(b, {h}, nullptr);
        for (auto& seq : plan.out_blocks) {
            TP skip = hs.back(); hs.pop_back();
            if (skip->d != h->d || skip->h != h->h || skip->w != h->w) fail("pixie_unet_forward: skip tensor and decoder tensor differ in size");
            std::vector<TP> parts{h, skip};   // th.cat([h, hs.pop()], dim=1), :932 -- never materialised
            h.reset(); skip.reset();
            for (const Blk& b : seq) {
                TP nh = block(b, parts, (b.kind == UP && !hs.empty()) ? hs.back().get() : nullptr);
                parts.clear();
                parts.push_back(nh);
            }
            h = parts[0];
        }
        stats(h);
        AB pro = norm_finalize(h->sums, h->c, h->spatial(), 0, 1, nullptr, nullptr);
        ConvOpt oo; oo.pro = &pro; oo.affine_store = "unet.out.0"; oo.act = ACT_LEAKY; oo.bound = norm_bound("unet.out.0", h->spatial());
        oo.out_ptr = d_out;
        return conv({h}, "unet.out.2", c.out_channels, 3, oo);
    }
};

Tens::~Tens() {
    if (!ex) return;
    if (off >= 0) ex->arena.release(off);
    if (sums_off >= 0) ex->arena.release(sums_off);
}

// max|weight| and max|bias| of every normalisation layer whose parameters changed: device reductions, ONE copy, ONE
// synchronisation (first pass after set_param only; a pass whose bounds are current never synchronises).
void refresh_bounds(pixie_unet* net, void* stream) {
    std::vector<std::pair<std::string, int>> todo;   // (layer key, word index)
    int n_norm = 0;
    for (const Param& p : net->params) {
        if (!p.norm || p.key.size() < 7 || p.key.compare(p.key.size() - 7, 7, ".weight") != 0) continue;
        const std::string key = p.key.substr(0, p.key.size() - 7);
        const Param& pb = net->param(key + ".bias");
        Bound& b = net->bounds[key];
        if (b.wver != p.version || b.bver != pb.version) todo.push_back({key, 2 * n_norm});
        ++n_norm;
    }
    if (todo.empty()) return;
    const size_t bytes = (size_t)2 * n_norm * sizeof(uint32_t);
    if (!net->d_bound_slots && hipMalloc(&net->d_bound_slots, bytes) != hipSuccess) fail("pixie_unet: hipMalloc failed");
    net->h_bound_slots.assign((size_t)2 * n_norm, 0u);
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(net->d_bound_slots, 0, bytes, s) != hipSuccess) fail("pixie_unet: hipMemsetAsync failed");
    for (auto& t : todo) {
        const Param& pw = net->param(t.first + ".weight");
        const Param& pb = net->param(t.first + ".bias");
        if (!pw.d || !pb.d) fail("pixie_unet_forward: parameter '%s' was never set", t.first.c_str());
        ok(pixie_tensor_amax(pw.d, pw.numel, net->d_bound_slots + t.second, stream), "pixie_tensor_amax");
        ok(pixie_tensor_amax(pb.d, pb.numel, net->d_bound_slots + t.second + 1, stream), "pixie_tensor_amax");
    }
    if (hipMemcpyAsync(net->h_bound_slots.data(), net->d_bound_slots, bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        fail("pixie_unet: reading the normalisation bounds back failed: %s", hipGetErrorString(hipGetLastError()));
    for (auto& t : todo) {
        Bound& b = net->bounds[t.first];
        std::memcpy(&b.wmax, &net->h_bound_slots[t.second], 4);
        std::memcpy(&b.bmax, &net->h_bound_slots[t.second + 1], 4);
        b.wver = net->param(t.first + ".weight").version;
        b.bver = net->param(t.first + ".bias").version;
    }
}

__global__ void zero_head_kernel(uint4* p, int64_t n16) {   // head_bytes is a multiple of kSlotAlign (256)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

constexpr int64_t kSlotAlign = 256;
int64_t slot_region_bytes(int n_slots) { return ((int64_t)n_slots * 4 + kSlotAlign - 1) / kSlotAlign * kSlotAlign; }
int64_t zsum_region_bytes(int64_t n_doubles) { return (n_doubles * 8 + kSlotAlign - 1) / kSlotAlign * kSlotAlign; }

Sized size_pass(pixie_unet* net, int D, int H, int W, bool from_proj0) {
    auto key = std::make_tuple(D, H, W, from_proj0);
    auto it = net->sized.find(key);
    if (it != net->sized.end()) return it->second;
    Exec ex;
    ex.net = net; ex.dry = true; ex.stream = nullptr;
    ex.arena.reset(reinterpret_cast<char*>(0x10000000), (int64_t)1 << 50);   // addresses of the dry run are never dereferenced
    {
        const float* fake = reinterpret_cast<const float*>(0x100);
        TP out = ex.forward(from_proj0 ? nullptr : fake, from_proj0 ? fake : nullptr, D, H, W, reinterpret_cast<float*>(0x100));
    }
    Sized r{ex.arena.peak + slot_region_bytes(ex.slot_next) + zsum_region_bytes(ex.zsum_next), ex.slot_next, ex.zsum_next};
    net->sized[key] = r;
    return r;
}

template <class F> int guarded(F f) {
    try { return f(); }
    catch (const std::exception& e) { return set_error("%s", e.what()); }
}

}  // namespace
}  // namespace pixie

extern "C" int pixie_unet_create(pixie_unet** out, const pixie_unet_config* c) {
    PX_REQUIRE(out && c, "pixie_unet_create: null argument");
    PX_REQUIRE(c->n_channel_mult >= 1 && c->n_channel_mult <= 8 && c->n_attention_resolutions >= 0 && c->n_attention_resolutions <= 8,
               "pixie_unet_create: channel_mult needs 1..8 entries, attention_resolutions 0..8");
    PX_REQUIRE(c->feature_channels > 0 && c->cond_dim > 0 && c->model_channels > 0 && c->num_res_blocks > 0 && c->grid_size > 0 && c->out_channels > 0,
               "pixie_unet_create: non-positive size");
    PX_REQUIRE(c->precision == 0 || c->precision == 1, "pixie_unet_create: precision is 0 (f16x3) or 1 (exact fp32)");
    return guarded([&] {
        std::unique_ptr<pixie_unet> n(new pixie_unet);
        n->cfg.feature_channels = c->feature_channels; n->cfg.cond_dim = c->cond_dim; n->cfg.model_channels = c->model_channels;
        n->cfg.num_res_blocks = c->num_res_blocks; n->cfg.grid_size = c->grid_size; n->cfg.out_channels = c->out_channels;
        n->cfg.precision = c->precision;
        n->cfg.mult.assign(c->channel_mult, c->channel_mult + c->n_channel_mult);
        n->cfg.attn.assign(c->attention_resolutions, c->attention_resolutions + c->n_attention_resolutions);
        n->plan = build_plan(n->cfg);
        n->params = param_table(n->cfg, n->plan);
        for (size_t i = 0; i < n->params.size(); ++i) n->index[n->params[i].key] = (int)i;
        const char* fs = getenv("PIXIE_FUSE_STATS"); n->fuse_stats = !(fs && fs[0] == '0');
        const char* sk = getenv("PIXIE_CONV_SPLIT_K"); n->split_k = !(sk && sk[0] == '0');
        const char* fk = getenv("PIXIE_FOLD_SKIP"); n->fold_skip = !(fk && fk[0] == '0');
        *out = n.release();
        return 0;
    });
}

extern "C" int pixie_unet_destroy(pixie_unet* h) {
    if (!h) return 0;
    for (auto& kv : h->packed16) if (kv.second.d) (void)hipFree(kv.second.d);
    for (auto& kv : h->packed32) if (kv.second.d) (void)hipFree(kv.second.d);
    if (h->d_bound_slots) (void)hipFree(h->d_bound_slots);
    while (!h->replays.empty()) h->drop_replay(0);
    delete h;
    return 0;
}

extern "C" int pixie_unet_param_count(const pixie_unet* h) { return h ? (int)h->params.size() : 0; }

extern "C" int pixie_unet_param_info(const pixie_unet* h, int i, const char** key, int64_t* numel, int32_t* ndim, int64_t shape[5]) {
    PX_REQUIRE(h && i >= 0 && i < (int)h->params.size(), "pixie_unet_param_info: index out of range");
    const Param& p = h->params[i];
    if (key) *key = p.key.c_str();
    if (numel) *numel = p.numel;
    if (ndim) *ndim = (int32_t)p.shape.size();
    if (shape) for (size_t k = 0; k < p.shape.size() && k < 5; ++k) shape[k] = p.shape[k];
    return 0;
}

extern "C" int pixie_unet_set_param(pixie_unet* h, const char* key, const float* d_values, int64_t numel) {
    PX_REQUIRE(h && key && d_values, "pixie_unet_set_param: null argument");
    auto it = h->index.find(key);
    PX_REQUIRE(it != h->index.end(), "pixie_unet_set_param: unexpected key '%s' (the reference's load_state_dict(strict=True) raises here too)", key);
    Param& p = h->params[it->second];
    PX_REQUIRE(numel == p.numel, "pixie_unet_set_param: '%s' has %lld elements, got %lld", key, (long long)p.numel, (long long)numel);
    p.d = d_values;
    ++p.version;
    ++h->epoch;
    return 0;
}

extern "C" int64_t pixie_unet_workspace_bytes(pixie_unet* h, int d, int hh, int w) {
    if (!h || d <= 0 || hh <= 0 || w <= 0) { set_error("pixie_unet_workspace_bytes: bad argument"); return -1; }
    int64_t bytes = -1;
    const int rc = guarded([&] {   // enough for either entry: from the feature grid, or behind an externally computed projector.net[0]
        bytes = size_pass(h, d, hh, w, false).bytes;
        if (h->cfg.projector_hidden() > 0) bytes = std::max(bytes, size_pass(h, d, hh, w, true).bytes);
        return 0;
    });
    return rc == 0 ? bytes : -1;
}

extern "C" int pixie_unet_forward(pixie_unet* h, const float* d_feat, const float* d_proj0, int d, int hh, int w, float* d_out,
                                  void* d_workspace, int64_t workspace_bytes, void* stream) {
    PX_REQUIRE(h && d_out && d_workspace && (d_feat || d_proj0), "pixie_unet_forward: null argument");
    PX_REQUIRE(d == h->cfg.grid_size && hh == d && w == d,
               "pixie_unet_forward: the network was built for a %d^3 grid (its LayerNorm parameters have that shape), got %d x %d x %d", h->cfg.grid_size, d, hh, w);
    PX_REQUIRE(!d_proj0 || h->cfg.projector_hidden() > 0, "pixie_unet_forward: d_proj0 needs the hidden-128 projector (feature_channels > cond_dim)");
    PX_REQUIRE(d_feat || (d_proj0 && h->cfg.has_projector()), "pixie_unet_forward: d_feat is null");
    return guarded([&] {
        const auto sized = size_pass(h, d, hh, w, d_proj0 != nullptr);
        if (workspace_bytes < sized.bytes)
            fail("pixie_unet_forward: workspace of %lld bytes, this grid needs %lld (pixie_unet_workspace_bytes)", (long long)workspace_bytes, (long long)sized.bytes);
        auto launch_all = [&] {
            Exec ex;
            ex.net = h; ex.dry = false; ex.stream = stream;
            const int64_t slot_bytes = slot_region_bytes(sized.slots), head_bytes = slot_bytes + zsum_region_bytes(sized.zsum_doubles);
            ex.slots = static_cast<uint32_t*>(d_workspace);
            ex.slot_cap = sized.slots;
            ex.zsums = reinterpret_cast<double*>(static_cast<char*>(d_workspace) + slot_bytes);
            ex.zsum_cap = sized.zsum_doubles;
            // Zeroed by a KERNEL, not hipMemsetAsync: as a node of a captured graph the memset did not reliably complete before
            // the kernels that follow it on ROCm 7.2 (replays of the f16x3 pass, whose first kernel already adds into this
            // region, gave wrong results at random once the workspace held a previous replay's values:
            // scripts/unet_soak.py, profiles/r3i_graph_replay_bisect.txt).  A kernel node has ordinary dependencies.
            hipLaunchKernelGGL(zero_head_kernel, dim3((unsigned)((head_bytes / 16 + 255) / 256)), dim3(256), 0, as_stream(stream),
                               static_cast<uint4*>(d_workspace), head_bytes / 16);
            if (hipGetLastError() != hipSuccess) fail("pixie_unet_forward: clearing the workspace head failed");
            ex.arena.reset(static_cast<char*>(d_workspace) + head_bytes, workspace_bytes - head_bytes);
            TP out = ex.forward(d_feat, d_proj0, d, hh, w, d_out);
        };
        if (h->use_graph) {
            for (size_t i = 0; i < h->replays.size(); ++i) {
                const pixie_unet::Replay& r = h->replays[i];
                if (r.feat != d_feat || r.proj0 != d_proj0 || r.out != d_out || r.ws != d_workspace || r.d != d) continue;
                if (r.epoch != h->epoch) { h->drop_replay(i); break; }   // parameters changed since the capture
                if (hipGraphLaunch(r.exec, as_stream(stream)) != hipSuccess) fail("pixie_unet_forward: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
                if (i + 1 != h->replays.size()) std::rotate(h->replays.begin() + (long)i, h->replays.begin() + (long)i + 1, h->replays.end());
                return 0;
            }
        }
        // Recording needs a capturable stream: refuse BEFORE doing the work, so that the call either succeeds completely or
        // fails without having produced a result the caller would have to throw away.
        if (h->use_graph && as_stream(stream) == nullptr)
            fail("pixie_unet_forward: option \"graph\" needs a non-default stream (the legacy stream cannot be captured); pass a created stream or set graph = 0");
        refresh_bounds(h, stream);
        launch_all();               // eager: packs what needs packing, and IS this call's result
        if (h->use_graph) {
            // the same launch sequence once more, recorded instead of executed, for the next call with these pointers
            hipStream_t st = as_stream(stream);
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess)
                fail("pixie_unet_forward: cannot capture on this stream (graph replay needs a non-default stream): %s", hipGetErrorString(hipGetLastError()));
            hipGraph_t graph = nullptr;
            try { launch_all(); }
            catch (...) { (void)hipStreamEndCapture(st, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
            if (hipStreamEndCapture(st, &graph) != hipSuccess || !graph) fail("pixie_unet_forward: hipStreamEndCapture failed: %s", hipGetErrorString(hipGetLastError()));
            hipGraphExec_t exec = nullptr;
            if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(graph); fail("pixie_unet_forward: hipGraphInstantiate failed"); }
            if (h->replays.size() >= pixie_unet::kMaxReplays) h->drop_replay(0);
            h->replays.push_back(pixie_unet::Replay{d_feat, d_proj0, d_out, d_workspace, d, h->epoch, graph, exec});
        }
        return 0;
    });
}

extern "C" int pixie_unet_set_option(pixie_unet* h, const char* key, int value) {
    PX_REQUIRE(h && key, "pixie_unet_set_option: null argument");
    const std::string k(key);
    if (k == "graph") {
        h->use_graph = value != 0;
        if (!h->use_graph) while (!h->replays.empty()) h->drop_replay(0);
        return 0;
    }
    return set_error("pixie_unet_set_option: unknown key '%s' (known: graph)", key);
}

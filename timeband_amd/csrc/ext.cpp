// The thin PyTorch-ROCm C++ extension over the C-ABI (include/timeband_hip.h): torch::autograd::Functions for the three
// differentiable ops that make up almost every launch of the training iteration -- one LSTM layer, Linear (+ activation),
// the single-output head -- so that an op is ONE call from Python into C++ instead of ~20 ctypes / allocator / stream
// look-ups from an autograd.Function written in Python (timeband_amd/ops.py keeps those as the reference
// implementation; tests run both).  Host code only: no kernels here, every launch goes through libtimeband_hip.so.
//
// Reference call sites replaced: source/utils/lstm_layer.py:44-46,83 (nn.LSTM), :49-50,84-87 (nn.Linear + Tanh /
// Sigmoid) and the autograd of those ops behind source/trainer.py:157,179,193.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPException.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "timeband_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;
using OptT = std::optional<Tensor>;

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", tb_last_error()); }

const float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fpm(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
Tensor opt(const OptT& t) { return t.has_value() ? *t : Tensor(); }

Tensor f32c(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "timeband_amd ops run on the GPU only (got a CPU tensor for ", what, "); there is no CPU fallback in the product path");
  TORCH_CHECK(t.scalar_type() == at::kFloat, "expected float32 for ", what);
  return t.is_contiguous() ? t : t.contiguous();
}

// ---- per-device runtime: sticky error words (shared with ops.Runtime), one growable workspace per stream, and the
// co-residency guard of the clustered recurrent launches (see ops.Runtime.clustered_begin for the rule) -------------
struct Clustered {
  uint64_t seq;
  bool excl;
  c10::hip::HIPStream stream;
};
struct Rt {
  Tensor err;
  std::unordered_map<void*, Tensor> ws;
  std::unordered_map<void*, Clustered> clustered;
  uint64_t seq = 0;
};
std::mutex g_mu;
std::unordered_map<int, Rt> g_rt;

Rt& rt_of(int dev) {
  auto it = g_rt.find(dev);
  TORCH_CHECK(it != g_rt.end() && it->second.err.defined(), "timeband_amd extension: no runtime for device ", dev,
              " (ops.Runtime.get registers it)");
  return it->second;
}

void* cur_stream(int dev) { return (void*)c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev).stream(); }

Tensor workspace(int dev, size_t nbytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rt& rt = rt_of(dev);
  void* key = cur_stream(dev);
  Tensor& ws = rt.ws[key];
  if (!ws.defined() || (size_t)ws.numel() < nbytes)
    ws = at::empty({(int64_t)(nbytes + nbytes / 4 + 256)}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev));
  return ws;
}

int* err_ptr(int dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  return rt_of(dev).err.data_ptr<int>();
}

// The guard WITHOUT an event per launch (the Python guard records one behind every clustered launch: ~16 per iteration,
// each a marker packet between two dependent kernels).  A stream's entry only remembers the order and kind of its last
// clustered launch; an event is recorded -- at that stream's TAIL, i.e. behind whatever was queued there since: a later
// point than the launch itself, so waiting for it is conservative, never too little -- only when another stream really
// has to wait: the third stream with a launch in flight, or either side of an exclusive launch.  With the default
// schedule (one side stream; the exclusive launches run while the other stream has been joined) no event is recorded
// at all.  A stream with nothing left to do (hipStreamQuery) drops out.
void wait_for_tail(c10::hip::HIPStream& other, c10::hip::HIPStream& cur) {
  hipEvent_t ev;
  C10_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  C10_HIP_CHECK(hipEventRecord(ev, other.stream()));
  C10_HIP_CHECK(hipStreamWaitEvent(cur.stream(), ev, 0));
  C10_HIP_CHECK(hipEventDestroy(ev));      // released once it has completed
}

void clustered_begin(int dev, bool exclusive) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rt& rt = rt_of(dev);
  if (rt.clustered.empty()) return;
  auto cur = c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev);
  void* key = (void*)cur.stream();
  // hipStreamQuery costs tens of microseconds on a busy stream: asked only where a wait would otherwise be queued
  std::vector<void*> live;
  for (auto it = rt.clustered.begin(); it != rt.clustered.end();) {
    if (it->first == key) { ++it; continue; }
    if (it->second.excl || exclusive) {
      if (it->second.stream.query()) { it = rt.clustered.erase(it); continue; }
      wait_for_tail(it->second.stream, cur);
    } else live.push_back(it->first);
    ++it;
  }
  if (live.size() > 1) {
    std::sort(live.begin(), live.end(), [&](void* a, void* b) { return rt.clustered.at(a).seq < rt.clustered.at(b).seq; });
    for (size_t i = 0; i + 1 < live.size(); ++i) {
      Clustered& c = rt.clustered.at(live[i]);
      if (c.stream.query()) rt.clustered.erase(live[i]);
      else wait_for_tail(c.stream, cur);
    }
  }
}

void clustered_end(int dev, bool exclusive) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rt& rt = rt_of(dev);
  auto cur = c10::hip::getCurrentHIPStream((c10::DeviceIndex)dev);
  void* key = (void*)cur.stream();
  auto it = rt.clustered.find(key);
  if (it == rt.clustered.end()) rt.clustered.emplace(key, Clustered{++rt.seq, exclusive, cur});
  else { it->second.seq = ++rt.seq; it->second.excl = exclusive; }
}

void set_runtime(int dev, Tensor err) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_rt[dev].err = err;
}

// ---- data parallel: an armed dp.GradReducer hears which parameters' gradients have just been enqueued (ops.grads_enqueued)
py::object* g_hook = nullptr;
std::atomic<bool> g_hook_armed{false};

void set_grads_hook(py::object hook) {
  if (!g_hook) g_hook = new py::object();
  *g_hook = hook;
}
void arm_grads_hook(bool on) { g_hook_armed.store(on); }

void notify(const std::vector<Tensor>& slots, const std::vector<Tensor>& params) {
  if (!g_hook_armed.load() || !g_hook) return;
  py::gil_scoped_acquire gil;
  py::list s, p;
  for (size_t i = 0; i < params.size(); ++i) {
    s.append(slots[i].defined() ? py::cast(slots[i]) : py::none());
    p.append(params[i].defined() ? py::cast(params[i]) : py::none());
  }
  (*g_hook)(s, p);
}

// ---- one LSTM layer --------------------------------------------------------------------------------------------
struct LstmOut {
  Tensor y, gates, cst;
};

LstmOut lstm_fwd_raw(const Tensor& x_, const Tensor& w_ih_, const Tensor& w_hh_, const Tensor& b_ih_, const Tensor& b_hh_,
                     bool train, int64_t split) {
  Tensor x = f32c(x_, "x"), w_ih = f32c(w_ih_, "w_ih"), w_hh = f32c(w_hh_, "w_hh"), b_ih = f32c(b_ih_, "b_ih"), b_hh = f32c(b_hh_, "b_hh");
  TORCH_CHECK(x.dim() == 3, "lstm: x must be (B, T, in)");
  const int B = (int)x.size(0), T = (int)x.size(1), In = (int)x.size(2), H = (int)w_hh.size(1);
  TORCH_CHECK(w_ih.size(0) == 4 * H && w_ih.size(1) == In && w_hh.size(0) == 4 * H, "lstm: weight shapes do not match the input");
  TORCH_CHECK(tb_lstm_supported(H), "lstm: hidden size ", H, " is not supported by the HIP kernels (supported: 16, 32, 64, 128, 256, 512)");
  const int dev = x.get_device();
  auto o = x.options();
  LstmOut r;
  r.y = at::empty({B, T, H}, o);
  r.gates = at::empty({B, T, 4 * H}, o);
  if (train) r.cst = at::empty({B, T, H}, o);
  const int step = split <= 0 ? B : (int)std::min<int64_t>(split, B);
  for (int b0 = 0; b0 < B; b0 += step) {
    const int nb = std::min(step, B - b0);
    Tensor ws = workspace(dev, tb_lstm_workspace_bytes(nb, T, In, H, 0));
    const bool excl = H > 64 && tb_lstm_exclusive(nb, In, H, 0);
    if (H > 64) clustered_begin(dev, excl);
    check(tb_lstm_fwd(fp(x) + (size_t)b0 * T * In, fp(w_ih), fp(w_hh), fp(b_ih), fp(b_hh), fpm(r.y) + (size_t)b0 * T * H,
                      fpm(r.gates) + (size_t)b0 * T * 4 * H, train ? fpm(r.cst) + (size_t)b0 * T * H : nullptr, nb, T, In, H,
                      train ? 1 : 0, ws.data_ptr(), (size_t)ws.numel(), err_ptr(dev), cur_stream(dev)),
          "tb_lstm_fwd");
    if (H > 64) clustered_end(dev, excl);
  }
  return r;
}

struct LSTMLayerFn : public torch::autograd::Function<LSTMLayerFn> {
  // args: x, w_ih, w_hh, b_ih, b_hh | train, dx_from, split | slots (4, optional) | pre y, gates, cst (optional)
  static Tensor forward(AutogradContext* ctx, Tensor x, Tensor w_ih, Tensor w_hh, Tensor b_ih, Tensor b_hh, bool train,
                        int64_t dx_from, int64_t split, OptT s_wih, OptT s_whh, OptT s_bih, OptT s_bhh, OptT pre_y,
                        OptT pre_gates, OptT pre_cst) {
    Tensor xc = f32c(x, "x"), wi = f32c(w_ih, "w_ih"), wh = f32c(w_hh, "w_hh");
    const int64_t B = xc.size(0), T = xc.size(1), H = wh.size(1);
    Tensor y, gates, cst;
    if (pre_y.has_value()) {
      y = pre_y->alias();
      gates = *pre_gates;
      cst = opt(pre_cst);
      TORCH_CHECK(y.sizes() == at::IntArrayRef({B, T, H}) && gates.sizes() == at::IntArrayRef({B, T, 4 * H}) &&
                      y.is_contiguous() && gates.is_contiguous(),
                  "lstm: precomputed buffers do not match the input");
    } else {
      LstmOut r = lstm_fwd_raw(xc, wi, wh, b_ih, b_hh, train, split);
      y = r.y; gates = r.gates; cst = r.cst;
    }
    if (train) {
      ctx->save_for_backward({xc, wi, wh, y, gates, cst, w_ih, w_hh, b_ih, b_hh, opt(s_wih), opt(s_whh), opt(s_bih), opt(s_bhh)});
      ctx->saved_data["dx_from"] = dx_from;
      ctx->saved_data["need"] = std::vector<bool>{x.requires_grad(), w_ih.requires_grad(), w_hh.requires_grad(),
                                                  b_ih.requires_grad(), b_hh.requires_grad()};
      ctx->saved_data["used"] = false;
    }
    return y;
  }

  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    TORCH_CHECK(!ctx->saved_data["used"].toBool(), "lstm: backward called twice (the gate stash is consumed in place)");
    ctx->saved_data["used"] = true;
    auto sv = ctx->get_saved_variables();
    Tensor x = sv[0], w_ih = sv[1], w_hh = sv[2], y = sv[3], gates = sv[4], cst = sv[5];
    std::vector<Tensor> params{sv[6], sv[7], sv[8], sv[9]};
    std::vector<Tensor> slots{sv[10], sv[11], sv[12], sv[13]};
    auto need = ctx->saved_data["need"].toBoolList();
    const int B = (int)x.size(0), T = (int)x.size(1), In = (int)x.size(2), H = (int)w_hh.size(1);
    const int dev = x.get_device();
    Tensor dy = f32c(grads[0], "dy");
    Tensor dx = need[0] ? at::empty_like(x) : Tensor();
    bool use_slots = false, slots_ok = true;
    for (int i = 0; i < 4; ++i) {
      if (need[i + 1]) {
        if (slots[i].defined()) use_slots = true;
        else slots_ok = false;
      } else slots[i] = Tensor();
    }
    use_slots = use_slots && slots_ok;
    Tensor dwi, dwh, dbi, dbh;
    if (use_slots) { dwi = slots[0]; dwh = slots[1]; dbi = slots[2]; dbh = slots[3]; }
    else {
      auto o = x.options();
      if (need[1]) dwi = at::empty({4 * H, In}, o);
      if (need[2]) dwh = at::empty({4 * H, H}, o);
      if (need[3]) dbi = at::empty({4 * H}, o);
      if (need[4]) dbh = at::empty({4 * H}, o);
    }
    Tensor ws = workspace(dev, tb_lstm_workspace_bytes(B, T, In, H, 1));
    const bool excl = H > 64 && tb_lstm_exclusive(B, In, H, 1);
    if (H > 64) clustered_begin(dev, excl);
    int64_t dxf = ctx->saved_data["dx_from"].toInt();
    dxf = std::min<int64_t>(std::max<int64_t>(dxf, 0), B);
    check(tb_lstm_bwd_rows(fp(dy), fp(x), fp(w_ih), fp(w_hh), fp(y), fpm(gates), fp(cst), fpm(dx), fpm(dwi), fpm(dwh), fpm(dbi),
                           fpm(dbh), B, T, In, H, use_slots ? 1 : 0, ws.data_ptr(), (size_t)ws.numel(), err_ptr(dev),
                           cur_stream(dev), (int)dxf),
          "tb_lstm_bwd_rows");
    if (H > 64) clustered_end(dev, excl);
    tensor_list out(15);
    out[0] = dx;
    if (use_slots) notify(slots, params);
    else { out[1] = dwi; out[2] = dwh; out[3] = dbi; out[4] = dbh; }
    return out;
  }
};

Tensor lstm_layer(Tensor x, Tensor w_ih, Tensor w_hh, Tensor b_ih, Tensor b_hh, int64_t dx_from, int64_t split, OptT s_wih,
                  OptT s_whh, OptT s_bih, OptT s_bhh, OptT pre_y, OptT pre_gates, OptT pre_cst) {
  // grad mode is off inside Function::forward, so the stash decision is taken here
  const bool train = at::GradMode::is_enabled() && (x.requires_grad() || w_ih.requires_grad() || w_hh.requires_grad() ||
                                                    b_ih.requires_grad() || b_hh.requires_grad());
  return LSTMLayerFn::apply(x, w_ih, w_hh, b_ih, b_hh, train, dx_from, split, s_wih, s_whh, s_bih, s_bhh, pre_y, pre_gates, pre_cst);
}

std::vector<Tensor> lstm_fwd_raw_py(Tensor x, Tensor w_ih, Tensor w_hh, Tensor b_ih, Tensor b_hh, bool train, int64_t split) {
  LstmOut r = lstm_fwd_raw(x, w_ih, w_hh, b_ih, b_hh, train, split);
  return {r.y, r.gates, r.cst};
}

// ---- Linear (+ activation) ---------------------------------------------------------------------------------------
Tensor linear_act_raw(const Tensor& x_, const Tensor& W_, const Tensor& b_, int64_t act) {
  Tensor x = f32c(x_, "x"), W = f32c(W_, "W");
  Tensor b = b_.defined() ? f32c(b_, "b") : Tensor();
  TORCH_CHECK(x.dim() == 2 && W.dim() == 2 && W.size(1) == x.size(1), "linear: weight shape does not match the input");
  const int M = (int)x.size(0), K = (int)x.size(1), N = (int)W.size(0);
  Tensor y = at::empty({M, N}, x.options());
  check(tb_linear_act_fwd(fp(x), fp(W), fp(b), fpm(y), M, K, N, (int)act, cur_stream(x.get_device())), "tb_linear_act_fwd");
  return y;
}

struct LinearActFn : public torch::autograd::Function<LinearActFn> {
  static Tensor forward(AutogradContext* ctx, Tensor x, Tensor W, OptT b, int64_t act, OptT s_W, OptT s_b, OptT pre) {
    Tensor xc = f32c(x, "x"), Wc = f32c(W, "W");
    Tensor y;
    if (pre.has_value()) {
      TORCH_CHECK(pre->dim() == 2 && pre->size(0) == xc.size(0) && pre->size(1) == Wc.size(0) && pre->is_contiguous(),
                  "linear: precomputed output does not match the input");
      y = pre->alias();
    } else {
      y = linear_act_raw(xc, Wc, opt(b), act);
    }
    ctx->save_for_backward({xc, Wc, act ? y : Tensor(), W, opt(b), opt(s_W), opt(s_b)});
    ctx->saved_data["act"] = act;
    ctx->saved_data["need"] = std::vector<bool>{x.requires_grad(), W.requires_grad(), b.has_value() && b->requires_grad()};
    return y;
  }

  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto sv = ctx->get_saved_variables();
    Tensor x = sv[0], W = sv[1], y = sv[2];
    std::vector<Tensor> params{sv[3], sv[4]};
    std::vector<Tensor> slots{sv[5], sv[6]};
    auto need = ctx->saved_data["need"].toBoolList();
    const int64_t act = ctx->saved_data["act"].toInt();
    const int M = (int)x.size(0), K = (int)x.size(1), N = (int)W.size(0);
    const int dev = x.get_device();
    void* st = cur_stream(dev);
    Tensor dy = f32c(grads[0], "dy");
    Tensor dz = dy;
    if (act) {
      dz = at::empty_like(dy);
      check(tb_act_bwd(fp(dy), fp(y), fpm(dz), (long)dy.numel(), (int)act, st), "tb_act_bwd");
    }
    Tensor dx = need[0] ? at::empty_like(x) : Tensor();
    bool use_slots = false, slots_ok = true;
    for (int i = 0; i < 2; ++i) {
      if (need[i + 1]) {
        if (slots[i].defined()) use_slots = true;
        else slots_ok = false;
      } else slots[i] = Tensor();
    }
    use_slots = use_slots && slots_ok;
    Tensor dW, db;
    if (use_slots) { dW = slots[0]; db = slots[1]; }
    else {
      if (need[1]) dW = at::empty_like(W);
      if (need[2]) db = at::empty({N}, x.options());
    }
    Tensor ws = workspace(dev, tb_linear_workspace_bytes(M, K, N));
    check(tb_linear_bwd(fp(dz), fp(x), fp(W), fpm(dx), fpm(dW), fpm(db), M, K, N, use_slots ? 1 : 0, ws.data_ptr(),
                        (size_t)ws.numel(), st),
          "tb_linear_bwd");
    tensor_list out(7);
    out[0] = dx;
    if (use_slots) notify(slots, params);
    else { out[1] = dW; out[2] = db; }
    return out;
  }
};

// ---- the discriminator's single-output head: act(x @ w.T + b), streaming kernels -----------------------------------
struct HeadFn : public torch::autograd::Function<HeadFn> {
  static Tensor forward(AutogradContext* ctx, Tensor x, Tensor W, OptT b, int64_t act, OptT s_W, OptT s_b) {
    Tensor xc = f32c(x, "x"), Wc = f32c(W, "W");
    Tensor bc = b.has_value() ? f32c(*b, "b") : Tensor();
    const long M = xc.size(0);
    const int K = (int)xc.size(1);
    Tensor y = at::empty({M, 1}, xc.options());
    check(tb_head_fwd(fp(xc), fp(Wc), fp(bc), fpm(y), M, K, (int)act, cur_stream(xc.get_device())), "tb_head_fwd");
    ctx->save_for_backward({xc, Wc, y, W, opt(b), opt(s_W), opt(s_b)});
    ctx->saved_data["act"] = act;
    ctx->saved_data["need"] = std::vector<bool>{x.requires_grad(), W.requires_grad(), b.has_value() && b->requires_grad()};
    return y;
  }

  static tensor_list backward(AutogradContext* ctx, tensor_list grads) {
    auto sv = ctx->get_saved_variables();
    Tensor x = sv[0], W = sv[1], y = sv[2];
    std::vector<Tensor> params{sv[3], sv[4]};
    std::vector<Tensor> slots{sv[5], sv[6]};
    auto need = ctx->saved_data["need"].toBoolList();
    const int64_t act = ctx->saved_data["act"].toInt();
    const long M = x.size(0);
    const int K = (int)x.size(1);
    const int dev = x.get_device();
    Tensor dy = f32c(grads[0], "dy");
    Tensor dx = need[0] ? at::empty_like(x) : Tensor();
    bool use_slots = false, slots_ok = true;
    for (int i = 0; i < 2; ++i) {
      if (need[i + 1]) {
        if (slots[i].defined()) use_slots = true;
        else slots_ok = false;
      } else slots[i] = Tensor();
    }
    use_slots = use_slots && slots_ok;
    Tensor dW, db;
    if (use_slots) { dW = slots[0]; db = slots[1]; }
    else {
      if (need[1]) dW = at::empty_like(W);
      if (need[2]) db = at::empty({1}, x.options());
    }
    Tensor ws = workspace(dev, tb_head_workspace_bytes(M, K));
    check(tb_head_bwd(fp(dy), fp(y), fp(x), fp(W), fpm(dx), fpm(dW), fpm(db), M, K, (int)act, use_slots ? 1 : 0, ws.data_ptr(),
                      (size_t)ws.numel(), cur_stream(dev)),
          "tb_head_bwd");
    tensor_list out(6);
    out[0] = dx;
    if (use_slots) notify(slots, params);
    else { out[1] = dW; out[2] = db; }
    return out;
  }
};

bool aligned16(const Tensor& t) { return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0; }

Tensor linear_act(Tensor x, Tensor W, OptT b, int64_t act, OptT s_W, OptT s_b, OptT pre) {
  if (!pre.has_value() && W.size(0) == 1 && tb_head_supported((int)W.size(1)) && x.is_contiguous() && W.is_contiguous() &&
      x.scalar_type() == at::kFloat && aligned16(x) && aligned16(W))
    return HeadFn::apply(x, W, b, act, s_W, s_b);
  return LinearActFn::apply(x, W, b, act, s_W, s_b, pre);
}

Tensor linear_act_raw_py(Tensor x, Tensor W, OptT b, int64_t act) { return linear_act_raw(x, W, opt(b), act); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "timeband_amd: autograd Functions over libtimeband_hip.so (C-ABI include/timeband_hip.h)";
  m.def("set_runtime", &set_runtime);
  m.def("clustered_begin", &clustered_begin);
  m.def("clustered_end", &clustered_end);
  m.def("set_grads_hook", &set_grads_hook);
  m.def("arm_grads_hook", &arm_grads_hook);
  m.def("lstm_layer", &lstm_layer, py::arg("x"), py::arg("w_ih"), py::arg("w_hh"), py::arg("b_ih"), py::arg("b_hh"),
        py::arg("dx_from") = 0, py::arg("split") = 0, py::arg("s_wih") = py::none(), py::arg("s_whh") = py::none(),
        py::arg("s_bih") = py::none(), py::arg("s_bhh") = py::none(), py::arg("pre_y") = py::none(),
        py::arg("pre_gates") = py::none(), py::arg("pre_cst") = py::none());
  m.def("lstm_fwd_raw", &lstm_fwd_raw_py);
  m.def("linear_act", &linear_act, py::arg("x"), py::arg("W"), py::arg("b"), py::arg("act") = 0, py::arg("s_W") = py::none(),
        py::arg("s_b") = py::none(), py::arg("pre") = py::none());
  m.def("linear_act_raw", &linear_act_raw_py);
}

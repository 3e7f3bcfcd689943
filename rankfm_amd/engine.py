"""Device-resident training session: the model and the interaction shard stay in HBM between calls.

`_rankfm._fit` (the drop-in boundary) uploads, trains and downloads on every call, which is what the reference's
call site expects.  Long jobs, the benchmark and the multi-GPU trainer instead keep everything resident as
torch tensors (PyTorch is used for device memory and streams only) and call `rfm_fit_device` on raw pointers.
"""
import ctypes as C

import numpy as np
import torch

from . import _hip

WEIGHT_NAMES = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")


class DeviceSession:
    RESERVE_EPOCHS = 64

    def __init__(self, interactions, sample_weight, csr_offsets, csr_items, x_uf, x_if, weights, *, alpha=0.01, beta=0.1,
                 learning_rate=0.1, learning_schedule="constant", learning_exponent=0.25, max_samples=1,
                 mode="hogwild", rng="counter", seed=1492, device=None, n_workgroups=0, rows_per_launch=0,
                 check_finite=True, want_penalty=False, has_user_features=None, has_item_features=None,
                 shape_override=0, hogwild_damping=0.0, debug_flags=0, tune=None, keep_layout=False):
        if not torch.cuda.is_available():
            raise _hip.EngineUnavailable("no MI355X visible to PyTorch-ROCm: rankfm_amd has no CPU fallback")
        _hip.lib()
        if device is None:
            device = torch.cuda.current_device()
        self.device = device if isinstance(device, torch.device) else torch.device("cuda", int(device))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if learning_schedule not in ("constant", "invscaling"):
            raise ValueError("unknown [learning_schedule]")
        dev = self.device

        def up(a, dtype):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=dtype).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)

        self.interactions = up(interactions, torch.int32)
        self.sample_weight = up(sample_weight, torch.float32)
        self.csr_offsets = up(csr_offsets, torch.int64)
        self.csr_items = up(csr_items, torch.int32)
        self.x_uf = up(x_uf, torch.float32)
        self.x_if = up(x_if, torch.float32)
        self.weights = {k: up(weights[k], torch.float32) for k in WEIGHT_NAMES}
        self.n_interactions = int(self.interactions.shape[0])
        self.n_users, self.n_factors = (int(s) for s in self.weights["v_u"].shape)
        self.n_items = int(self.weights["v_i"].shape[0])
        self.n_user_features = int(self.weights["v_uf"].shape[0])
        self.n_item_features = int(self.weights["v_if"].shape[0])
        self.has_uf = int(bool((self.x_uf != 0).any().item())) if has_user_features is None else int(has_user_features)
        self.has_if = int(bool((self.x_if != 0).any().item())) if has_item_features is None else int(has_item_features)
        self.hyper = dict(alpha=alpha, beta=beta, learning_rate=learning_rate,
                          learning_schedule=_hip.SCHEDULE_CONSTANT if learning_schedule == "constant" else _hip.SCHEDULE_INVSCALING,
                          learning_exponent=learning_exponent, max_samples=int(max_samples))
        self.mode = _hip.MODE_SERIAL if mode == "serial" else _hip.MODE_HOGWILD
        self.rng = _hip.RNG_MT19937 if rng == "mt19937" else _hip.RNG_COUNTER
        self.seed = int(seed) & 0xFFFFFFFF
        self.n_workgroups, self.rows_per_launch = int(n_workgroups), int(rows_per_launch)
        self.check_finite, self.want_penalty = int(check_finite), int(want_penalty)
        self._workspace = None
        self.shape_override = int(shape_override)
        self.hogwild_damping = float(hogwild_damping)
        self._plan_token = 0
        self.debug_flags = int(debug_flags)
        # the experiments' overrides (rfm_fit_tuning): part of the plan, so fixed per session; None = production (a NULL pointer)
        self._tuning = _hip.make_tuning(tune, n_workgroups=self.n_workgroups, rows_per_launch=self.rows_per_launch,
                                        debug_shape=self.shape_override, debug_flags=self.debug_flags)
        self._geometry = None
        # keep_layout: between `run` calls the item-side weights stay in the engine's working layout inside the workspace (segment-major
        # factor rows, padded biases: rfm_fit_config.keep_layout) -- `self.weights["v_i"]` / `["w_i"]` are then STALE until `sync_weights()`
        # (which `weights_to_host`, `predict` and `recommend` call).  For a resident single-GPU training loop; callers that touch the
        # weight tensors between runs themselves (the multi-GPU exchange works on them) leave it off.
        self.keep_layout = bool(keep_layout)
        self._layout_token = 0
        self._layout_cfg = None

    def _config(self, epochs, epoch_begin, part=None, rng_epoch_offset=0):
        cfg = _hip.FitConfig(
            rng_epoch_offset=int(rng_epoch_offset),
            epoch_part_index=part[0] if part else 0, epoch_parts=part[1] if part else 0,
            hogwild_damping=self.hogwild_damping, plan_token=int(self._plan_token),
            keep_layout=int(self.keep_layout), layout_token=int(self._layout_token),
            n_interactions=self.n_interactions, n_users=self.n_users, n_items=self.n_items,
            n_user_features=self.n_user_features, n_item_features=self.n_item_features, n_factors=self.n_factors,
            has_user_features=self.has_uf, has_item_features=self.has_if,
            epochs=int(epochs), epoch_begin=int(epoch_begin), mode=self.mode, rng=self.rng, seed=self.seed,
            check_finite=self.check_finite, want_penalty=self.want_penalty, **self.hyper)
        if self._tuning is not None:
            cfg.tuning = C.pointer(self._tuning)
        return cfg

    def _buffers(self, perms_t=None):
        w = self.weights
        return _hip.FitBuffers(
            interactions=self.interactions.data_ptr(), sample_weight=self.sample_weight.data_ptr(),
            csr_offsets=self.csr_offsets.data_ptr(), csr_items=self.csr_items.data_ptr(),
            x_uf=self.x_uf.data_ptr(), x_if=self.x_if.data_ptr(),
            w_i=w["w_i"].data_ptr(), w_if=w["w_if"].data_ptr(), v_u=w["v_u"].data_ptr(), v_i=w["v_i"].data_ptr(),
            v_uf=w["v_uf"].data_ptr(), v_if=w["v_if"].data_ptr(),
            perms=perms_t.data_ptr() if perms_t is not None else None,
            workspace=self._workspace.data_ptr(), workspace_bytes=self._workspace.numel())

    def sync_weights(self):
        """with `keep_layout`: bring `self.weights["v_i"]` / `["w_i"]` up to date from the engine's working layout in the workspace
        (rfm_fit_export_weights; enqueued on the current stream, no host synchronisation).  The workspace stays current: training goes on
        from it.  A no-op when nothing was kept."""
        if not self._layout_token:
            return
        cfg = self._layout_cfg
        cfg.plan_token, cfg.layout_token = int(self._plan_token), int(self._layout_token)
        buf = self._buffers()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _hip.raise_for_status(_hip.lib().rfm_fit_export_weights(C.byref(cfg), C.byref(buf), C.c_void_p(stream)))

    def invalidate_layout(self):
        """the caller has written `self.weights["v_i"]` / `["w_i"]` itself: export what the engine holds first (`sync_weights`), then call
        this so that the next run imports the tensors again"""
        self._layout_token = 0

    def run(self, epochs=1, epoch_begin=0, perms=None, raise_on_error=True, part=None, rng_epoch_offset=0):
        """train `epochs` epochs in place on the resident tensors; returns the per-epoch report (numpy arrays)"""
        cfg = self._config(epochs, epoch_begin, part, rng_epoch_offset)    # part = (k, n): only the k-th of n slices of each epoch's order
        # (the workspace is sized for calls of up to RESERVE_EPOCHS epochs from the start: its per-epoch arrays are a few hundred bytes an
        #  epoch, and a longer call than the first would otherwise re-allocate it and plan again)
        size_cfg = self._config(max(int(epochs), self.RESERVE_EPOCHS), epoch_begin, part, rng_epoch_offset)
        need = _hip.lib().rfm_fit_workspace_bytes(C.byref(size_cfg))
        if need == 0:
            _hip.raise_for_status(_hip.lib().rfm_fit_supported(C.byref(cfg)))
        if self._workspace is None or self._workspace.numel() < need:
            self.sync_weights()                  # (a kept layout lives in the workspace that is about to be replaced)
            self._workspace = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            self._plan_token = self._layout_token = 0
            cfg.plan_token = cfg.layout_token = 0
        if perms is not None and self._layout_token:
            self.sync_weights()                  # (explicit orders run another kernel on another plan: back to the caller's layout first)
            self._layout_token = cfg.layout_token = 0
        perms_t = None
        if perms is not None:
            perms_t = torch.as_tensor(np.ascontiguousarray(perms, dtype=np.int32)).to(self.device)
            assert tuple(perms_t.shape) == (epochs, self.n_interactions)
        buf = self._buffers(perms_t)
        ll = np.zeros(epochs, dtype=np.float64)
        pen = np.zeros(epochs, dtype=np.float64)
        ms = np.zeros(epochs, dtype=np.float32)
        draws = np.zeros(epochs, dtype=np.int64)
        rep = _hip.FitReport(
            log_likelihood=ll.ctypes.data_as(C.POINTER(C.c_double)), reg_penalty=pen.ctypes.data_as(C.POINTER(C.c_double)),
            sgd_kernel_ms=ms.ctypes.data_as(C.POINTER(C.c_float)), n_draws=draws.ctypes.data_as(C.POINTER(C.c_int64)))
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = _hip.lib().rfm_fit_device(C.byref(cfg), C.byref(buf), C.c_void_p(stream), C.byref(rep))
        if rc in (_hip.ERR_BAD_ARG, _hip.ERR_UNKNOWN_SCHEDULE, _hip.ERR_UNSUPPORTED, _hip.ERR_WORKSPACE, _hip.ERR_NO_DEVICE):
            pass                                 # (refused before anything ran: plan and layout in the workspace are what they were)
        else:
            self._plan_token = int(rep.plan_token) if (rc == _hip.OK and perms is None) else 0
            self._layout_token = int(rep.layout_token) if self._plan_token else 0
            self._layout_cfg = cfg if self._layout_token else None
        self._geometry = dict(rep.geometry(), single_group=bool(self.debug_flags & 1), seed=self.seed,
                              epoch_part=part)
        out = dict(status=rc, log_likelihood=ll, reg_penalty=pen, sgd_kernel_ms=ms, n_draws=draws,
                   epochs_done=rep.epochs_done, launches_per_epoch=rep.launches_per_epoch,
                   waves_per_launch=rep.waves_per_launch)
        if raise_on_error:
            _hip.raise_for_status(rc)
        return out

    def geometry(self):
        """launch geometry of the last run (what rankfm_amd.order needs to mirror the engine's draws on the host)"""
        return self._geometry

    def step_scales(self):
        """The Hogwild step damping of the last run's plan as two arrays: the scale of the positive item's step per item [I] and of
        the user's step per user [U] (both 1 where nothing is damped).  Diagnostic: the sequential oracle can apply the same scales
        (its `pos_step` / `user_step`), which separates what the damping changes -- a deliberate change of the optimiser -- from what
        asynchronous execution changes.  Read from the plan at the head of the workspace (rfm_api.hip `carve`: pos_scale [I] comes
        first, a hot slot s is encoded as scale + 2 (s + 1)) and the launch geometry (user_cap = M x segments / interactions in
        flight, rfm_api.hip "plan, part 3")."""
        g = self._geometry
        m = 128.0 if self.hogwild_damping == 0 else self.hogwild_damping
        if self._workspace is None or g is None or m <= 0 or g.get("single_group") or self.mode == _hip.MODE_SERIAL:
            return np.ones(self.n_items, np.float32), np.ones(self.n_users, np.float32)
        raw = self._workspace[:4 * self.n_items].view(torch.float32).cpu().numpy().astype(np.float64)
        slot = np.where(raw >= 2.0, np.floor(raw * 0.5), 0.0)
        pos = (raw - 2.0 * slot).astype(np.float32)
        user_cap = m * float(g["n_units"]) / float(g["working_groups"])
        deg = np.maximum(np.diff(self.csr_offsets.cpu().numpy()), 1)
        return pos, np.minimum(1.0, user_cap / deg).astype(np.float32)

    def weights_to_host(self):
        self.sync_weights()
        return {k: v.detach().cpu().numpy() for k, v in self.weights.items()}

    # ---- serving on the resident model: `_predict` / `_recommend` (rankfm/_rankfm.pyx:345-390, 393-460) without the uploads of the
    #      host entry points -- the model, the feature matrices and the users' item lists are already in HBM --------------------------------
    def _model_view(self):
        self.sync_weights()
        w = self.weights
        return _hip.ModelView(
            n_users=self.n_users, n_items=self.n_items, n_user_features=self.n_user_features, n_item_features=self.n_item_features,
            n_factors=self.n_factors, has_user_features=self.has_uf, has_item_features=self.has_if,
            x_uf=self.x_uf.data_ptr(), x_if=self.x_if.data_ptr(), w_i=w["w_i"].data_ptr(), w_if=w["w_if"].data_ptr(),
            v_u=w["v_u"].data_ptr(), v_i=w["v_i"].data_ptr(), v_uf=w["v_uf"].data_ptr(), v_if=w["v_if"].data_ptr())

    def _to_device(self, a, ndim, what):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if t.dim() != ndim:
            raise ValueError("[%s] must have %d dimension(s)" % (what, ndim))
        return t

    def predict(self, pairs, to_host=True):
        """float32 [n, 2] (user, item) index pairs (NaN = unknown id) -> float32 [n] scores, NaN where either index is NaN"""
        pairs_d = self._to_device(pairs, 2, "pairs")
        scores = torch.empty(pairs_d.shape[0], dtype=torch.float32, device=self.device)
        mv = self._model_view()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = _hip.lib().rfm_predict_device(C.byref(mv), pairs_d.shape[0], C.c_void_p(pairs_d.data_ptr()), C.c_void_p(scores.data_ptr()),
                                               C.c_void_p(stream))
        _hip.raise_for_status(rc)
        return scores.cpu().numpy() if to_host else scores

    def recommend(self, users, n_items=10, filter_previous=False, to_host=True):
        """float32 [n] user indexes (NaN = unknown) -> float32 [n, n_items] item indexes by descending utility (NaN rows for unknown
        users); `filter_previous` skips the items of the session's own CSR lists"""
        users_d = self._to_device(users, 1, "users")
        n_items = int(n_items)
        if n_items < 1 or n_items > self.n_items:
            raise ValueError("[n_items] must be between 1 and the number of training items")
        n = int(users_d.shape[0])
        rec = torch.empty((n, n_items), dtype=torch.float32, device=self.device)
        mv = self._model_view()
        need = int(_hip.lib().rfm_recommend_workspace_bytes(C.byref(mv), n, n_items))
        ws = getattr(self, "_serve_workspace", None)
        if ws is None or ws.numel() < need:
            ws = self._serve_workspace = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = _hip.lib().rfm_recommend_device(C.byref(mv), n, C.c_void_p(users_d.data_ptr()), C.c_void_p(self.csr_offsets.data_ptr()),
                                                 C.c_void_p(self.csr_items.data_ptr()), n_items, int(bool(filter_previous)),
                                                 C.c_void_p(rec.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream))
        _hip.raise_for_status(rc)
        return rec.cpu().numpy() if to_host else rec

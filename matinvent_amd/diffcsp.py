"""DiffCSP diffusion module on the HIP path.

Mirror of models/diffcsp/diffusion.py:69-399 (DiffCSPModule): same constructor hparams,
same buffers (`beta_scheduler.*`, `sigma_scheduler.*`), same methods
`sample / add_noise / calc_sample_loss / calc_kl_reg`, same `decoder.*` parameter names in
`state_dict()`.  The arithmetic runs in libmatinvent_hip.so; there is no PyTorch fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .cspnet import CSPNet, CrystalBatch, MAX_ATOMIC_NUM, _ptr, _stream
from .schedules import BetaScheduler, SigmaScheduler, sampler_coefficients, time_embedding_freqs


def _scatter_mean(src, index, dim_size):
    """torch_scatter.scatter(..., reduce='mean') on the device: sum / max(count, 1)."""
    out = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add(0, index, src)
    cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device).index_add(0, index, torch.ones_like(src))
    return out / cnt.clamp(min=1)


def _cfg(d, drop=("_target_",)):
    return {k: v for k, v in dict(d).items() if k not in drop}


class SinusoidalTimeEmbeddings(nn.Module):
    """diffusion.py:53-66, evaluated on the device from the host-built frequency table."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.register_buffer("freqs", time_embedding_freqs(dim), persistent=False)

    def forward(self, time):
        lib = _lib.load()
        times = time.to(self.freqs.device, torch.int32).contiguous()
        out = torch.empty(times.shape[0], self.dim, device=self.freqs.device)
        _lib.check(lib.mi_time_embedding(_ptr(times), _ptr(self.freqs), times.shape[0], self.dim, _ptr(out), _stream()))
        return out


class _Counts:
    """Atom counts of a contiguous group of crystals (what `sample` needs of a batch)."""

    def __init__(self, num_atoms):
        self.num_atoms = torch.as_tensor(num_atoms, dtype=torch.long)


class DiffCSPModule(nn.Module):
    def __init__(self, decoder, beta_scheduler, sigma_scheduler, latent_dim=0, time_dim=256, cost_lattice=1.0,
                 cost_coord=1.0, cost_type=20.0, device=None, **kwargs):
        super().__init__()
        self.hparams = dict(decoder=dict(decoder), beta_scheduler=dict(beta_scheduler), sigma_scheduler=dict(sigma_scheduler),
                            latent_dim=latent_dim, time_dim=time_dim, cost_lattice=cost_lattice, cost_coord=cost_coord,
                            cost_type=cost_type, **kwargs)
        dev = torch.device(device if device is not None else "cuda")
        # diffusion.py:73 -- latent_dim + time_dim, pred_type=True, smooth=True
        self.decoder = CSPNet(**_cfg(decoder), latent_dim=latent_dim + time_dim, pred_type=True, smooth=True, device=dev)
        self.beta_scheduler = BetaScheduler(**_cfg(beta_scheduler))
        self.sigma_scheduler = SigmaScheduler(**_cfg(sigma_scheduler))
        self.time_dim = time_dim
        self.time_embedding = SinusoidalTimeEmbeddings(time_dim)
        self.cost_lattice, self.cost_coord, self.cost_type = cost_lattice, cost_coord, cost_type
        self.keep_lattice = cost_lattice < 1e-5   # diffusion.py:78-79: CSP mode, that part of the structure is given
        self.keep_coords = cost_coord < 1e-5
        self.to(dev)

    @property
    def device(self):
        return self.decoder.theta.device

    def _coefficients(self, step_lr):
        """Per-step scalar table, cached until a scheduler buffer changes (checkpoint load)."""
        key = (float(step_lr),) + tuple(b._version for b in (self.beta_scheduler.alphas, self.beta_scheduler.alphas_cumprod,
                                                              self.beta_scheduler.sigmas, self.sigma_scheduler.sigmas,
                                                              self.sigma_scheduler.sigmas_norm))
        cache = self.__dict__.setdefault("_coef_cache", {})
        if key not in cache:
            cache.clear()
            cache[key] = sampler_coefficients(self.beta_scheduler, self.sigma_scheduler, step_lr).contiguous()
        return cache[key]

    def crystal_batch(self, batch, node_offset=0, graph_offset=0) -> CrystalBatch:
        """Index tables + workspace for `batch` (anything with .num_atoms); cached on the object
        PER MODULE (agent and frozen prior must not share workspace: a forward of one would
        clobber the activations the other's pending backward reads)."""
        cache = getattr(batch, "_mi_batches", None)
        if cache is None:
            cache = {}
            try:
                batch._mi_batches = cache
            except AttributeError:
                pass
        key = (id(self), node_offset, graph_offset)
        cb = cache.get(key)
        if cb is None or cb.num_atoms_list != [int(x) for x in batch.num_atoms.tolist()]:
            cb = self.decoder.make_batch(batch.num_atoms, node_offset, graph_offset)
            cache[key] = cb
        return cb

    # ---- fine-tune surface (pipeline/mat_invent.py:152-161) ------------------------------------
    def add_noise(self, batch, time=None, noise=None, seed=None):
        """DiffCSPModule.add_noise (diffusion.py:81-119) for an explicit timestep index
        (`time` in 0..T-1 -> diffusion time T - time, :86-87).  Returns the reference's triple
        (noised_input, noises, batch.batch).  Fresh Gaussian noise per call from the Philox
        stream (seed, running call counter) unless `noise` = (rand_l, rand_x, rand_t)."""
        lib = _lib.load()
        dev = self.device
        T = self.beta_scheduler.timesteps
        cb = self._batch_for(batch.num_atoms)
        B, N = cb.num_graphs, cb.num_nodes
        sched = None
        if time is None:   # :83-84: one uniformly drawn time per crystal (numpy's global generator, like the reference)
            times = self.beta_scheduler.uniform_sample_t(B, dev)
            ac = self.beta_scheduler.alphas_cumprod[times]
            sched = torch.stack([torch.sqrt(ac), torch.sqrt(1.0 - ac), self.sigma_scheduler.sigmas[times],
                                 self.sigma_scheduler.sigmas_norm[times]], dim=1).to(torch.float32).contiguous()
        else:
            t = T - int(time)
            times = torch.full((B,), t, device=dev)
            ac = self.beta_scheduler.alphas_cumprod[t]
            c0, c1 = float(torch.sqrt(ac)), float(torch.sqrt(1.0 - ac))
            sig, sn = float(self.sigma_scheduler.sigmas[t]), float(self.sigma_scheduler.sigmas_norm[t])
        time_emb = self.time_embedding(times)
        f = lambda x: x.to(dev, torch.float32).contiguous()
        lengths, angles, frac0 = f(batch.lengths), f(batch.angles), f(batch.frac_coords)
        at = batch.atom_types.to(dev, torch.int32).contiguous()
        in_lat, in_frac = torch.empty(B, 3, 3, device=dev), torch.empty(N, 3, device=dev)
        in_types, tar_x = torch.empty(N, MAX_ATOMIC_NUM, device=dev), torch.empty(N, 3, device=dev)
        rand_l, rand_t = torch.empty(B, 3, 3, device=dev), torch.empty(N, MAX_ATOMIC_NUM, device=dev)
        nz = (None, None, None) if noise is None else tuple(f(x) for x in noise)
        self._noise_calls = getattr(self, "_noise_calls", 0) + 1
        seed = getattr(self, "noise_seed", 0) if seed is None else seed
        if sched is not None:
            _lib.check(lib.mi_add_noise_per_crystal(cb._h, _ptr(lengths), _ptr(angles), _ptr(frac0), _ptr(at), _ptr(sched), seed,
                                                    self._noise_calls & 0xFFFFFFFF, _ptr(nz[0]), _ptr(nz[1]), _ptr(nz[2]), _ptr(in_lat),
                                                    _ptr(in_frac), _ptr(in_types), _ptr(tar_x), _ptr(rand_l), _ptr(rand_t), _stream()),
                       "mi_add_noise_per_crystal")
        else:
            _lib.check(lib.mi_add_noise(cb._h, _ptr(lengths), _ptr(angles), _ptr(frac0), _ptr(at), c0, c1, sig, sn, seed,
                                        self._noise_calls & 0xFFFFFFFF, _ptr(nz[0]), _ptr(nz[1]), _ptr(nz[2]), _ptr(in_lat), _ptr(in_frac),
                                        _ptr(in_types), _ptr(tar_x), _ptr(rand_l), _ptr(rand_t), _stream()), "mi_add_noise")
        noised_input = (time_emb, in_types, in_frac, in_lat, cb.num_atoms, cb.batch)
        return noised_input, (rand_l, tar_x, rand_t), cb.batch

    def predict(self, input_all):
        """The network evaluation of calc_sample_loss alone (diffusion.py:124-125): library kernels only, no torch arithmetic -- the part
        that may run on a side stream beside another network's forward (finetune._ft_step_module_surface)."""
        noised_input, _, node2graph = input_all
        time_emb, atom_types, frac, lattices, num_atoms, _ = noised_input
        return self.decoder(time_emb, atom_types, frac, lattices, num_atoms, node2graph, batch=self._batch_for(num_atoms))

    def calc_sample_loss(self, input_all, pred=None):
        """diffusion.py:121-138: per-crystal cost_lattice*mse_l + cost_coord*mse_x + cost_type*mse_t (`pred`: the output of `predict`,
        when the caller has run the network already)."""
        noised_input, (rand_l, tar_x, rand_t), node2graph = input_all
        time_emb, atom_types, frac, lattices, num_atoms, _ = noised_input
        B = lattices.shape[0]
        pred_l, pred_x, pred_t = pred if pred is not None else self.predict(input_all)
        loss_lattice = torch.pow(pred_l - rand_l, 2).mean(dim=(1, 2))
        loss_coord = _scatter_mean(torch.pow(pred_x - tar_x, 2).mean(dim=1), node2graph, B)
        loss_type = _scatter_mean(torch.pow(pred_t - rand_t, 2).mean(dim=1), node2graph, B)
        loss = self.cost_lattice * loss_lattice + self.cost_coord * loss_coord + self.cost_type * loss_type
        return loss, (pred_l, pred_x, pred_t)

    def calc_kl_reg(self, agent_pred, prior_pred, batch):
        """diffusion.py:140-149: per-crystal sum of three mean-squared agent-prior differences."""
        pl, px, pt = agent_pred
        plp, pxp, ptp = (p.detach() for p in prior_pred)
        node2graph = batch.batch if hasattr(batch, "batch") else batch
        node2graph = node2graph.to(pl.device)
        B = pl.shape[0]
        k0 = torch.pow(pl - plp, 2).mean(dim=(1, 2))
        k1 = _scatter_mean(torch.pow(px - pxp, 2).mean(dim=1), node2graph, B)
        k2 = _scatter_mean(torch.pow(pt - ptp, 2).mean(dim=1), node2graph, B)
        return k0 + k1 + k2

    def forward(self, noised_input):
        time_emb, atom_types, frac, lattices, num_atoms, node2graph = noised_input
        return self.decoder(time_emb, atom_types, frac, lattices, num_atoms, node2graph, batch=self._batch_for(num_atoms))

    def _batch_for(self, num_atoms):
        """CrystalBatch of THIS module for a num_atoms tensor (small cache keyed by the atom counts and
        the shard offsets `self.shard_offsets` = (first global atom, first global crystal), which only
        enter the noise counters)."""
        off = getattr(self, "shard_offsets", (0, 0))
        key = (tuple(int(x) for x in num_atoms.tolist()), off)
        cache = self.__dict__.setdefault("_nb_cache", {})
        cb = cache.get(key)
        if cb is None:
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            cb = self.decoder.make_batch(list(key[0]), off[0], off[1])
            cache[key] = cb
        return cb

    @torch.no_grad()
    def sample(self, batch, diff_ratio=1.0, step_lr=1e-5, seed=0, noise=None, init=None, record=False, t_start=None,
               t_stop=0, node_offset=0, graph_offset=0, streams=None):
        """DiffCSPModule.sample (diffusion.py:273-399).

        `streams` > 1 splits the crystals into that many contiguous groups and runs their chains CONCURRENTLY on separate
        HIP streams (crystals never interact, and the counter-based noise is indexed by global atom / crystal id, so the
        samples are the same as those of the unsplit batch): the node-level kernels and the partial last round of one
        group's edge GEMMs overlap the other groups' edge GEMMs.  None = automatic (2-4 for large batches).
        """
        if self.__dict__.get("_knn_pending"):
            self.check_graph()   # (the verdict of the previous call's chains: by now they have long finished)
        if isinstance(batch, CrystalBatch):
            return self._sample_one(batch, step_lr, seed, noise, init, record, t_start, t_stop, node_offset, graph_offset)
        if (self.keep_lattice or self.keep_coords) and init is None:
            # CSP mode (diffusion.py:283-287): the known part of the structure replaces the drawn initial state and is never moved
            cb0 = self.crystal_batch(batch, node_offset, graph_offset)
            dev = self.device
            x0, l0, a0 = (torch.empty(cb0.num_nodes, 3, device=dev), torch.empty(cb0.num_graphs, 3, 3, device=dev),
                          torch.empty(cb0.num_nodes, MAX_ATOMIC_NUM, device=dev))
            _lib.check(_lib.load().mi_sampler_init_state(cb0._h, seed, self.beta_scheduler.timesteps, _ptr(a0), _ptr(x0), _ptr(l0), _stream()))
            if self.keep_coords:
                x0 = batch.frac_coords.to(dev, torch.float32)
            if self.keep_lattice:
                from .data import lattice_params_to_matrix
                l0 = lattice_params_to_matrix(batch.lengths.to(dev, torch.float32), batch.angles.to(dev, torch.float32))
            init = (x0, l0, a0)
        na = [int(v) for v in batch.num_atoms.tolist()]
        if streams is None:
            e_total = sum(v * v for v in na)
            # measured: 2 chains +8 % at E = 26k (192 mp_20-sized crystals); at E = 102k 1/2/3/4/5 chains give
            # 24.3 / 25.8 / 26.7 / 27.4 / 22.8 structures/s (the runtime has four hardware queues)
            streams = 4 if e_total >= 98304 else 3 if e_total >= 49152 else 2 if e_total >= 16384 else 1
        streams = max(1, min(int(streams), len(na)))
        if streams == 1:
            return self._sample_one(batch, step_lr, seed, noise, init, record, t_start, t_stop, node_offset, graph_offset)
        key = ("split", streams, tuple(na))
        parts = getattr(batch, "_mi_split", {}).get(key)
        if parts is None:  # contiguous crystal groups, cached on the batch object like its CrystalBatch
            cuts = [len(na) * k // streams for k in range(streams + 1)]
            parts = [_Counts(na[cuts[k]:cuts[k + 1]]) for k in range(streams)]
            try:
                batch._mi_split = {key: parts}
            except AttributeError:
                pass
        g0 = [0]
        n0 = [0]
        for p_ in parts:
            g0.append(g0[-1] + len(p_.num_atoms))
            n0.append(n0[-1] + int(p_.num_atoms.sum()))
        self.decoder.sync()
        self._coefficients(step_lr)
        cur = torch.cuda.current_stream()
        # ONE state for the whole batch, allocated (and, with `init`, copied) here on the caller's stream; every chain works in place on ITS rows
        # of it.  (Each chain used to clone its slices on its own thread -- twelve small torch calls contending for the interpreter lock at the
        # head of every call, 70-150 us apart on the GPU timeline -- and the call ended with five torch.cat launches: profiles/r5_window_steps.log.)
        dev = self.device
        n_tot, b_tot = n0[-1], g0[-1]
        if init is not None:
            state = tuple(v.to(dev, torch.float32).contiguous().clone() for v in init)
        else:
            state = (torch.empty(n_tot, 3, device=dev), torch.empty(b_tot, 3, 3, device=dev), torch.empty(n_tot, MAX_ATOMIC_NUM, device=dev))
        ready = cur.record_event()
        # long-lived worker threads, one per stream (streams.ChainWorkers): a call hands each its chain and waits -- no thread start per call
        from .streams import ChainWorkers
        workers = ChainWorkers.get(streams, self.device)

        # (starting chain k later by k x 150 ... 1 200 us was measured and costs the delay: DESIGN 19.7, profiles/r5_chain_stagger.log)
        def run(k, stream):
            stream.wait_event(ready)
            own = (state[0][n0[k]:n0[k + 1]], state[1][g0[k]:g0[k + 1]], state[2][n0[k]:n0[k + 1]])   # (contiguous row ranges: views, no copy)
            nz = None if noise is None else {"corr_x": noise["corr_x"][:, n0[k]:n0[k + 1]], "pred_x": noise["pred_x"][:, n0[k]:n0[k + 1]],
                                             "pred_t": noise["pred_t"][:, n0[k]:n0[k + 1]], "pred_l": noise["pred_l"][:, g0[k]:g0[k + 1]]}
            r = self._sample_one(parts[k], step_lr, seed, nz, None, record, t_start, t_stop, node_offset + n0[k], graph_offset + g0[k],
                                 inplace=own, drawn=init is not None)
            cur.wait_event(stream.record_event())
            return r

        out = workers.run(run, streams)

        def merge(ds):
            m = {}
            for d in ds:  # the parts were allocated on the group streams and are consumed on the caller's stream
                for v in d.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
            for name in ds[0]:
                if name == "batch_idx":
                    m[name] = torch.cat([d[name] + g0[k] for k, d in enumerate(ds)])
                else:
                    m[name] = torch.cat([d[name] for d in ds])
            return m

        def final_state():   # the chains' final states ARE the rows of `state`: nothing to concatenate
            idx = getattr(batch, "_mi_batch_idx", None)
            if idx is None or idx[0] != key:
                idx = (key, torch.cat([o[0]["batch_idx"] + g0[k] for k, o in enumerate(out)]), torch.cat([o[0]["num_atoms"] for o in out]))
                try:
                    batch._mi_batch_idx = idx
                except AttributeError:
                    pass
            return dict(atom_types=state[2], frac_coords=state[0], lattices=state[1], num_atoms=idx[2], batch_idx=idx[1])
        traj = {t: (final_state() if t == t_stop else merge([o[1][t] for o in out])) for t in out[0][1]}
        return traj[t_stop], traj

    def check_graph(self):
        """knn edge style: wait for the chains enqueued by `sample` and raise (MI_ECAPACITY, as a RuntimeError) if a neighbour list built inside one of them
        exceeded its capacity -- the results of that chain are invalid.  Called by `sample` for the chains of EARLIER calls, by DiffCSPSampler.generate before
        it unpacks a batch, and by any caller about to read a chain's results.  A no-op for the fc edge style."""
        pending, self.__dict__["_knn_pending"] = self.__dict__.get("_knn_pending", []), []
        lib = _lib.load()
        for cb, stream in pending:
            _lib.check(lib.mi_knn_graph_status(cb._h, C.c_void_p(stream.cuda_stream)), "mi_knn_graph_status")

    def _sample_one(self, batch, step_lr, seed, noise, init, record, t_start, t_stop, node_offset, graph_offset, inplace=None, drawn=False):
        """One chain over one CrystalBatch on the current stream.

        Returns (traj[t_stop], traj) like the reference.  `traj` holds every step only when
        record=True (the reference keeps all T+1 states alive on the device); otherwise it
        holds the final state.  Noise comes from the library's counter-based Philox stream
        keyed by `seed` unless `noise` supplies per-step arrays (dict with corr_x/pred_l/
        pred_t/pred_x: [T+1, ...] tensors) and `init` the initial state (x_T, l_T, t_T).
        """
        lib = _lib.load()
        dev = self.device
        T = self.beta_scheduler.timesteps
        t_start = T if t_start is None else t_start
        cb = batch if isinstance(batch, CrystalBatch) else self.crystal_batch(batch, node_offset, graph_offset)
        B, N = cb.num_graphs, cb.num_nodes
        self.decoder.sync()
        if inplace is not None:   # (a chain of a split batch: its rows of the caller's state, updated in place; `drawn`: they already hold the initial state)
            x, l, a = inplace
            assert x.is_contiguous() and l.is_contiguous() and a.is_contiguous() and x.shape[0] == N and l.shape[0] == B
            if not drawn:
                _lib.check(lib.mi_sampler_init_state(cb._h, seed, T, _ptr(a), _ptr(x), _ptr(l), _stream()), "mi_sampler_init_state")
        elif init is None:
            x = torch.empty(N, 3, device=dev)
            l = torch.empty(B, 3, 3, device=dev)
            a = torch.empty(N, MAX_ATOMIC_NUM, device=dev)
            _lib.check(lib.mi_sampler_init_state(cb._h, seed, T, _ptr(a), _ptr(x), _ptr(l), _stream()), "mi_sampler_init_state")
        else:
            x, l, a = (v.to(dev, torch.float32).contiguous().clone() for v in init)
        # (traj[T]['frac_coords'] = x_T % 1, diffusion.py:289: mi_sampler_run wraps the coordinates in place before its first step -- no torch
        #  arithmetic on a chain's stream)
        coef = self._coefficients(step_lr)
        _lib.check(lib.mi_sampler_set_keep(cb._h, int(self.keep_lattice), int(self.keep_coords)))
        nz = None
        if noise is not None:
            keep = {k: noise[k].to(dev, torch.float32).contiguous() for k in ("corr_x", "pred_l", "pred_t", "pred_x")}
            nz = _lib.SamplerNoise(*(keep[k].data_ptr() for k in ("corr_x", "pred_l", "pred_t", "pred_x")))
        rec, rec_t = None, None
        if record:
            z = lambda *s: torch.zeros(*s, device=dev)
            rec_t = dict(atom_types=z(T + 1, N, MAX_ATOMIC_NUM), frac_coords=z(T + 1, N, 3), lattices=z(T + 1, B, 3, 3),
                         frac_coords_mid=z(T + 1, N, 3), log_prob_l=z(T + 1, B), log_prob_t=z(T + 1, B), log_prob_x=z(T + 1, B))
            rec = _lib.SamplerRecord(*(rec_t[k].data_ptr() for k in ("atom_types", "frac_coords", "lattices", "frac_coords_mid",
                                                                     "log_prob_l", "log_prob_t", "log_prob_x")))
        _lib.check(lib.mi_sampler_run(self.decoder._h, cb._h, coef.numpy().ctypes.data_as(C.POINTER(C.c_float)), T, t_start, t_stop,
                                      _ptr(self.time_embedding.freqs), seed, C.byref(nz) if nz is not None else None,
                                      C.byref(rec) if rec is not None else None, _ptr(a), _ptr(x), _ptr(l), _stream()),
                   "mi_sampler_run")
        if cb.edge_style == "knn":
            # the chain rebuilt its periodic neighbour list in every evaluation WITHOUT a host round trip (the reference synchronises per evaluation:
            # cspnet.py:243-257); a list over capacity could not raise inside the enqueued chain -- its verdict waits on the batch handle for check_graph()
            self.__dict__.setdefault("_knn_pending", []).append((cb, torch.cuda.current_stream()))
        final = dict(atom_types=a, frac_coords=x, lattices=l, num_atoms=cb.num_atoms, batch_idx=cb.batch)
        traj = {t_stop: final}
        if record:
            for t in range(t_start, t_stop - 1, -1):
                d = dict(atom_types=rec_t["atom_types"][t], frac_coords=rec_t["frac_coords"][t], lattices=rec_t["lattices"][t],
                         num_atoms=cb.num_atoms, batch_idx=cb.batch)
                if t > max(t_stop, 1):
                    d.update(log_prob_l=rec_t["log_prob_l"][t], log_prob_t=rec_t["log_prob_t"][t],
                             log_prob_x=rec_t["log_prob_x"][t], frac_coords_mid=rec_t["frac_coords_mid"][t])
                traj[t] = d
        return final, traj

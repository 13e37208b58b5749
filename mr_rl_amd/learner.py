"""FusedLearner: the DDPG update of RL/MR_ddpg.py:288-305 as ONE launch of libmrsim.so (mrsim_ddpg_update, include/mrsim.h).

    agent = DDPG(env, fused=True)        # update() / update_graphed() / train_collected() then run this kernel

The online / target parameters, Adam moments and step counts live in flat device vectors in the library's documented layout;
the agent's nn.Modules (actor, actor_t, critic, critic_t) are re-pointed at VIEWS of those vectors, so everything that reads
the modules (the eager policy, DeviceActor.load_module_device, state_dict) sees the learner's current parameters without a copy.
Batch normalisation must be in the reference's mode (fixed moving statistics): DDPG(bn_mode="reference").
"""
import ctypes as C

from . import _lib

# (module attribute path, offset) in the library's parameter vector (mr_rl_amd/csrc/mrsim_learner.h)
ACTOR_LAYOUT = (("fc1.weight", 0), ("fc1.bias", 320), ("bn1.weight", 384), ("bn1.bias", 448), ("fc2.weight", 512), ("fc2.bias", 4608),
                ("bn2.weight", 4672), ("bn2.bias", 4736), ("out.weight", 4800), ("out.bias", 4928))
CRITIC_LAYOUT = (("fc1.weight", 4932), ("fc1.bias", 5252), ("bn1.weight", 5316), ("bn1.bias", 5380), ("t1.weight", 5444),
                 ("t2.weight", 7492), ("t2.bias", 7556), ("out.weight", 7588), ("out.bias", 7620))


def _get(module, path):
    obj = module
    for part in path.split("."):
        obj = getattr(obj, part)
    return obj


class FusedLearner:
    def __init__(self, agent):
        import torch
        self.agent = agent
        dev = agent.buffer.s.device
        if dev.type != "cuda":
            raise RuntimeError("FusedLearner needs a HIP device (libmrsim has no CPU fallback)")
        self._L = _lib.lib()
        P = _lib.DDPG_PARAMS
        self.online = torch.zeros(P, dtype=torch.float32, device=dev)
        self.target = torch.zeros(P, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(P, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(P, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(P, dtype=torch.float32, device=dev)
        self.steps = torch.zeros(2, dtype=torch.int32, device=dev)
        self.losses = torch.zeros(2, dtype=torch.float32, device=dev)
        self.seed, self.draws = 0x5EED0000 + 12345, 0     # in-kernel sampler: Philox key and the update counter it is keyed by
        self.idx_out = None                               # set to an int32 [batch] tensor to record the rows of the next updates
        self._alias(agent.actor, ACTOR_LAYOUT, self.online)
        self._alias(agent.critic, CRITIC_LAYOUT, self.online)
        self._alias(agent.actor_t, ACTOR_LAYOUT, self.target)
        self._alias(agent.critic_t, CRITIC_LAYOUT, self.target)
        stats = []
        for nets in ((agent.actor, agent.critic), (agent.actor_t, agent.critic_t)):
            for bn in (nets[0].bn1, nets[0].bn2, nets[1].bn1):
                stats += [bn.running_mean.detach().float(), bn.running_var.detach().float()]
        self.bn_stats = torch.stack(stats).contiguous()           # [2][3][2][64]: constants of the update (reference BN mode)
        bound = agent.actor.action_bound.detach().cpu().tolist()
        opt = agent.opt_a.defaults
        # work space of the multi-workgroup form (batches above 64 on batch / 64 compute units): sized for the largest batch, zeroed
        # once; multi_workgroup=False (a test / measurement switch) keeps the single workgroup looping over the tiles
        self.batch_scratch = torch.zeros(_lib.ddpg_batch_scratch_floats(_lib.DDPG_MAX_BATCH), dtype=torch.float32, device=dev)
        self.struct = _lib.MrsimDdpgLearner(
            self.online.data_ptr(), self.target.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.grad.data_ptr(),
            self.steps.data_ptr(), self.bn_stats.data_ptr(), float(agent.actor.bn1.eps), float(agent.gamma), float(agent.tau),
            float(agent.actor_lr), float(agent.critic_lr), float(opt["betas"][0]), float(opt["betas"][1]), float(opt["eps"]),
            (C.c_float * 2)(float(bound[0]), float(bound[1])), self.batch_scratch.data_ptr(), self.batch_scratch.numel())

    @property
    def multi_workgroup(self):
        return bool(self.struct.batch_scratch)

    @multi_workgroup.setter
    def multi_workgroup(self, on):
        self.struct.batch_scratch = self.batch_scratch.data_ptr() if on else None

    @staticmethod
    def _alias(module, layout, flat):
        import torch
        with torch.no_grad():
            for path, off in layout:
                p = _get(module, path)
                view = flat[off:off + p.numel()].view(p.shape)
                view.copy_(p.detach())
                p.data = view                                       # the module now reads and writes the flat vector

    def export_to_modules(self):
        """nothing to do: the modules alias the learner's vectors"""

    def update(self, batch=None, n=1, pack_into=None):
        """n consecutive updates in ONE launch (each draws its own rows when batch is None).  pack_into: a DeviceActor of one
        parameter block -- after the last update the online actor is folded and packed into that block by the same launch
        (MrsimDdpgLearner.actor_blob; the bits of DeviceActor.load_from_learner)."""
        if pack_into is None:
            return self._update(batch, n)
        st = self.struct
        st.actor_blob = pack_into.blobs[0].data_ptr()
        for j, x in enumerate(pack_into.weights["obs_scale"]):
            st.actor_obs_scale[j] = float(x)
        try:
            return self._update(batch, n)
        finally:
            st.actor_blob = None

    def _update(self, batch=None, n=1):
        import torch
        ag = self.agent
        buf = ag.buffer
        if batch is None:
            # the rows are drawn INSIDE the kernel (Philox keyed by (seed, update counter); without repetition up to 256 rows,
            # random.sample's law): the whole update is one launch, and the host only passes the ring's fill count.  The pointer
            # arguments never change: converted once (a call per env step is bound by the host otherwise)
            nb = ag.min_batch
            if nb % 64 != 0 or nb > _lib.DDPG_MAX_BATCH:
                raise ValueError("the fused learner takes batches that are multiples of 64 (<= %d)" % _lib.DDPG_MAX_BATCH)
            key = (buf.s.data_ptr(), None if self.idx_out is None else self.idx_out.data_ptr(), self.losses.data_ptr(), nb)
            if getattr(self, "_ring_key", None) != key:
                p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
                self._ring_args = (C.byref(self.struct), int(nb)), (p(buf.s), p(buf.a), p(buf.r), p(buf.t), p(buf.s2), None)
                self._ring_tail = (None if self.idx_out is None else p(self.idx_out), p(self.losses))
                self._ring_key = key
            head, ptrs = self._ring_args
            strm = C.c_void_p(torch.cuda.current_stream(buf.s.device).cuda_stream)
            _lib.check(self._L.mrsim_ddpg_update(*head, int(n), *ptrs, int(buf.size()), self.seed, self.draws, *self._ring_tail, strm),
                       "mrsim_ddpg_update")
            self.draws += int(n)
            return self.losses[0], self.losses[1]
        s, a, r, t, s2 = (x.contiguous().float() for x in batch)
        nb = s.shape[0]
        if nb % 64 != 0 or nb > _lib.DDPG_MAX_BATCH:
            raise ValueError("the fused learner takes batches that are multiples of 64 (<= %d)" % _lib.DDPG_MAX_BATCH)
        strm = C.c_void_p(torch.cuda.current_stream(buf.s.device).cuda_stream)
        p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        io = None if self.idx_out is None else p(self.idx_out)
        _lib.check(self._L.mrsim_ddpg_update(C.byref(self.struct), int(nb), int(n), p(s), p(a), p(r), p(t), p(s2), None, 0,
                                             self.seed, self.draws, io, p(self.losses), strm), "mrsim_ddpg_update")
        self.draws += int(n)
        self._keep = (s, a, r, t, s2)      # alive until the next call (the launch is asynchronous)
        return self.losses[0], self.losses[1]

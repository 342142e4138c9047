"""A train step captured in a HIP graph: forward + loss + backward + fused AdamW replayed as ONE graph launch.

Every entry point of include/equiformer_hip.h only enqueues work on the caller's stream (no allocation, no synchronisation),
so the ~280 launches of a QM9 step are legal inside a stream capture (tests/test_gpu_capture.py); replaying them removes the
host's launch path from the step -- ~0.4 ms of gaps at the head of an eager 10-ms step (profiles/r05/r05_fin_step_timeline.txt).
[ref: the train loop, engine.py:58-92; SURVEY.md 8d "HIP-graph captured where possible"]

What stays OUTSIDE the graph, and why:
  * the radius graph (EdgeGraph.from_radius): its edge count is the one data-dependent size of the step and is read back on
    the host.  The captured launches address the index tensors of the graph object the capture ran on; every later step
    rebuilds the radius graph INTO those tensors (`from_radius(into=...)`).  A step whose node / edge counts differ from the
    captured ones runs eagerly (and, from `min_eager` eager steps of a shape on, gets a graph of its own: one per shape).
  * three numbers the captured launches read from device words instead of frozen by-value arguments: the seed offset of
    the attention dropout (a fresh draw per replay: eqf_attn_aggregate_*_dseed) and {lr, 1 - b1^t, sqrt(1 - b2^t)} of AdamW
    (eqf_adamw_step_dev); the host writes them (pinned buffer, asynchronous copy) before it launches the graph.

The MD17 force-loss step (forces by a create_graph backward inside the forward, then a second-order backward) is captured the
same way (tests/test_gpu_capture.py; bench.py --workload md17_l2: 463 -> 619 frames/s, md17_l3: 218 -> 254).

Limits: one process / one GPU (a data-parallel reducer's collectives stay eager: `CapturedTrainStep` refuses a reducer),
attention dropout together with a create_graph backward (no model of the reference combines them), radius graphs only (the
periodic OC20 graph builds its by-source view with a device sort: eager), inputs other than the graph at fixed addresses
(`forward_loss` reads the same tensors every step; copy a new batch into them)."""
import torch

from . import ops


class CapturedTrainStep:
    """cs = CapturedTrainStep(model_params_owner_optimizer, forward_loss)
       loss = cs.step(build_graph)          # every train step

    forward_loss(graph) -> scalar loss, reading the batch from tensors that keep their addresses; build_graph(into) -> the
    EdgeGraph of this step's batch (`EdgeGraph.from_radius(pos, batch, r, into=into)`), called OUTSIDE the capture.
    optimizer: a FlatAdamW without a reducer."""

    def __init__(self, optimizer, forward_loss, min_eager=3, max_graphs=4):
        if getattr(optimizer, "_reducer", None) is not None:
            raise ValueError("CapturedTrainStep: data-parallel steps stay eager (the reducer's collectives are not captured)")
        self.opt, self.forward_loss = optimizer, forward_loss
        self.min_eager, self.max_graphs = int(min_eager), int(max_graphs)
        self._graphs = {}   # (N, E) -> dict(graph=CUDAGraph, sg=EdgeGraph, loss=Tensor)
        self._seen = {}     # (N, E) -> (eager steps so far, last EdgeGraph)
        dev = optimizer.flat_p.device
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._seed_host = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.replays = self.eager_steps = 0

    # ---- one eager step (also what a captured graph records) -------------------------------------------------------
    def _run(self, g):
        self.opt.zero_grad(set_to_none=True)
        loss = self.forward_loss(g)
        loss.backward()
        self.opt.step()
        # detached: a caller that keeps the loss of an EAGER step must not keep that step's autograd graph alive into a later
        # capture (hipStreamEndCapture crashed with one alive: tools/capture_debug4.py, round 6)
        return loss.detach()

    def _draw_seed(self):
        self._seed_host[0] = int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator, as the eager layers draw theirs
        self._seed_dev.copy_(self._seed_host, non_blocking=True)

    def step(self, build_graph):
        # the radius graph of this batch: into the tensors of a captured shape when one fits (tried most recent first)
        g = None
        for key, rec in reversed(list(self._graphs.items())):
            if not getattr(rec["sg"], "_radius_static", False):  # (abandoned by a build that found another edge count)
                del self._graphs[key]
                continue
            self._draw_seed()  # (the device words go out BEFORE the graph build's host read-back: they overlap it)
            g = build_graph(rec["sg"])
            if g is rec["sg"]:
                self.opt.advance_captured()
                rec["graph"].replay()
                self.replays += 1
                return rec["loss"]
            del self._graphs[key]  # its tensors were overwritten by the build of another shape: the graph is gone
            break  # (one rebuild attempt per step: a second one would repeat the neighbour search)
        if g is None:
            g = build_graph(None)
        key = (g.N, g.E)
        n, _ = self._seen.get(key, (0, None))
        if n < self.min_eager or len(self._graphs) >= self.max_graphs or not getattr(g, "_radius_static", False):
            self._seen[key] = (n + 1, g)
            self.eager_steps += 1
            return self._run(g)
        # capture this shape: the launches record the addresses of g's tensors, of the inputs forward_loss reads and of the
        # gradients / activations the graph's private pool hands out
        self.opt.device_hyper(True)
        graph = torch.cuda.CUDAGraph()
        with ops.dropout_seed_offset(self._seed_dev), ops._arena.capture_scope():
            step_before = self.opt._step
            with torch.cuda.graph(graph):
                loss = self._run(g)
            self.opt._step = step_before  # (capturing enqueued nothing: the step count advances with the replays)
        rec = dict(graph=graph, sg=g, loss=loss)
        self._graphs[key] = rec
        self._draw_seed()
        self.opt.advance_captured()
        graph.replay()
        self.replays += 1
        return loss

"""
Several batches in flight on one GPU: R independently captured hipGraphs of the same forward step,
replayed round-robin on R HIP streams.

Why: the joint step (enh_att.py forward: STFT -> LSTM masks -> MVDR -> conformer) is a chain of two
very different phases.  The LSTM mask estimator is bound by the hand-off latency between its
workgroups and leaves most of the chip idle; the conformer is bound by the matrix pipes.  One stream
runs them back to back; with two batches in flight the LSTM of one hides behind the GEMMs of the
other (MI355X, BASELINE configs[4], same box: 4.9 -> 3.9 ms per 32 utterances; the front-end
workload, a chain of short latency-bound launches, gains 30 % with three).

Every replica owns its buffers (a private graph memory pool per capture), so replays never alias.
The recurrent kernels synchronise their workgroups through memory and need all of them resident at
once: while a GraphReplicas object lives, the library state `nn_ops.lstm_share()` = R tells every
persistent launch of the process (captured or eager) to size its grid for 1 / R of the chip -- the
explicit `share` argument of aps_lstm_layer / aps_lstm_stack -- or to refuse the shape (the caller
then runs smaller chunks); two half-resident grids would otherwise wait on each other forever.
Eager launches issued while R replicas are in flight have to be ordered against them by the caller
(`submit(after_caller=True)` does); an oversubscribed chip shows up as a reported hand-off timeout
(nn_ops.lstm_timeouts, checked by synchronize()), never as silently wrong numbers.

Several streams are OPT-IN (the default is one).  Round 2 met a build-dependent disturbance between
kernels of different streams that share a CU (DESIGN.md "co-residency"): isolated wrong values in
the output of one kernel while a particular build of another ran beside it.  The shipped build does
not show it, but nothing in the stack promises that for the next compiler or driver, so with more than
one stream every `guard_every`-th submission is checked while the service runs: the graph just
launched beside the others is replayed once more ALONE on the same inputs and the two results must
agree bit for bit (RuntimeError otherwise; `checks_run` counts the checks).
"""
import contextlib
from typing import Any, Callable, List, Tuple

import torch as th

from . import _native, nn_ops


@contextlib.contextmanager
def concurrent_launches(n: int):
    """Launches issued inside size their memory-synchronised grids for n of them running at once
    (also the way to get an eager result that is bit-identical to a replica's)."""
    nn_ops.push_lstm_share(n)
    try:
        yield
    finally:
        nn_ops.pop_lstm_share(n)


# HIP multiplexes streams onto a handful of hardware queues (4 by default).  A process that keeps
# creating streams -- one set per GraphReplicas object -- ends up with two "concurrent" streams on ONE
# queue, i.e. serialised (measured: the second GraphReplicas of a process ran its two streams at the
# one-stream rate).  So the streams are process-wide, per device, and handed out again.
_STREAMS = {}


def replica_streams(device: th.device, n: int) -> List[th.cuda.Stream]:
    key = device.index if device.index is not None else th.cuda.current_device()
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(th.cuda.Stream(device=th.device("cuda", key)))
    return pool[:n]


class GraphReplicas:
    """
    fn: a no-argument callable launching one step on the current stream (inputs are whatever it
        closes over: static device tensors, refilled by the caller between submissions) -- or a
        LIST of such callables, one per resident input batch: every callable is captured into its
        own graph and the graphs are replayed round-robin (graph i on stream i % replicas), so a
        caller that rotates over P distinct input batches needs no copy into a static buffer
    replicas: batches in flight = streams (default 1 = everything on one stream; more is opt-in)
    verify: replay every graph a few times right after capture and compare with the eager step
        (bit-exact; the step must be deterministic), RuntimeError on a mismatch
    guard_every: with replicas > 1, every guard_every-th submit() replays the graph it launched once
        more with nothing else in flight and compares the two outputs bit for bit (None: 64 when
        replicas > 1; 0: off).  The step must read inputs the caller does not overwrite before the
        submit() call returns (it blocks for the check).

    The captured step must be IDEMPOTENT: `verify` and the guard replay a graph more than once per
    submission and compare outputs bit for bit, so a step with in-place state -- BatchNorm running
    statistics in train() mode, step counters, accumulators -- advances twice per checked submission and
    is reported as a disturbance.  Capture eval-mode forwards (what this class is for), or switch the
    checks off (`verify=False, guard_every=0`).  The default is one batch in flight (`replicas=1`); the
    two-stream mode bench.py's headline uses is opt-in.
    """

    def __init__(self, fn, replicas: int = 1, verify: bool = True, guard_every=None) -> None:
        if replicas < 1:
            raise ValueError(f"replicas must be >= 1, got {replicas}")
        if guard_every is None:
            guard_every = 64 if replicas > 1 else 0
        if guard_every < 0:
            raise ValueError(f"guard_every must be >= 0, got {guard_every}")
        self.guard_every = int(guard_every) if replicas > 1 else 0
        self.checks_run = 0
        self._submitted = 0
        _native.load()  # no HIP extension, no graphs: fail here, loudly
        fns: List[Callable[[], Any]] = list(fn) if isinstance(fn, (list, tuple)) else [fn] * replicas
        if len(fns) < replicas:
            raise ValueError(f"{len(fns)} step functions for {replicas} replicas")
        self.replicas = replicas
        self.streams: List[th.cuda.Stream] = []
        self.graphs: List[th.cuda.CUDAGraph] = []
        self.outputs: List[Any] = []
        self._next = 0
        self._holds_share = False
        nn_ops.push_lstm_share(replicas)  # held until close(): see the module docstring
        self._holds_share = True
        try:
            # Warm-up on the CALLER's stream, then a full stop, then the captures.  Round 1 found
            # replica 0 corrupted a few replays later when the warm-up ran on the capture stream
            # itself; round 2 root-caused it (scripts/memset_node_repro.py, no aps_amd kernel
            # involved): on ROCm 7.0 / 7.2 a hipMemsetAsync NODE recorded on a stream that still has
            # eager work queued in front of the capture stops executing from the third replay on,
            # so the LSTM's sentinel re-arm silently did nothing.  The launchers no longer record
            # memset nodes (fill kernels, csrc/common.h), which removes the cause; this order and
            # the self-check below stay as the defence against the same class of runtime bug.
            distinct = list(dict.fromkeys(fns))
            eager = {f: _clone(f()) for f in distinct}
            want = [eager[f] for f in fns]
            th.cuda.synchronize()
            self.streams = replica_streams(th.device("cuda", th.cuda.current_device()), replicas)
            for i, f in enumerate(fns):
                graph = th.cuda.CUDAGraph()
                with th.cuda.graph(graph, stream=self.streams[i % replicas],
                                   capture_error_mode="thread_local"):
                    out = f()
                self.graphs.append(graph)
                self.outputs.append(out)
        except BaseException:
            self.close()
            raise
        if verify:
            self._self_check(want)
        else:
            import warnings
            warnings.warn("GraphReplicas(verify=False): the post-capture self-check is the guard "
                          "against capture / replay bugs of the runtime (see the memset-node note)")

    def close(self) -> None:
        """give the chip back to full-size launches (idempotent; also called on collection)"""
        if self._holds_share:
            self._holds_share = False
            nn_ops.pop_lstm_share(self.replicas)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def __len__(self) -> int:
        return len(self.graphs)

    def check_outputs(self, want: List[Any], what: str = "") -> None:
        """every graph's current output equals want[i] bit for bit"""
        for i, out in enumerate(self.outputs):
            for a, b in zip(_leaves(out), _leaves(want[i])):
                if not th.equal(a, b):
                    raise RuntimeError(f"GraphReplicas: graph {i} differs from the eager step {what}")

    def _self_check(self, want: List[Any], rounds: int = 4) -> None:
        """every graph reproduces the eager step bit for bit, replay after replay"""
        dev = _leaves(want[0])[:1]
        for rnd in range(rounds):
            for _ in range(2 * len(self.graphs)):
                self.submit()
            self.synchronize()
            scratch = [th.empty(1 + 37 * rnd, device=r.device) for r in dev]  # allocator traffic
            self.check_outputs(want, f"after {rnd + 1} rounds of replays")
            del scratch
        self.eager_outputs = want

    def submit(self, after_caller: bool = True) -> Tuple[int, Any]:
        """Launch the next graph; returns (index, its output tensors).  The outputs are valid once
        that graph's stream has been waited on (wait(index) / synchronize()).
        after_caller: order the replay after the work already queued on the caller's stream, so
        inputs written there are visible (an event record + wait, ~10 us of host time: callers
        whose inputs do not change between submissions pass False)."""
        i = self._next
        self._next = (i + 1) % len(self.graphs)
        stream = self.streams[i % self.replicas]
        if after_caller:
            stream.wait_stream(th.cuda.current_stream())
        with th.cuda.stream(stream):
            self.graphs[i].replay()
        self._submitted += 1
        if self.guard_every and self._submitted % self.guard_every == 0:
            self._guard(i)
        return i, self.outputs[i]

    def _guard(self, i: int) -> None:
        """graph i was just launched beside whatever the other streams are running: its result must
        equal that of the same graph replayed with the chip to itself"""
        self.synchronize()
        beside = _clone(self.outputs[i])
        with th.cuda.stream(self.streams[i % self.replicas]):
            self.graphs[i].replay()
        self.streams[i % self.replicas].synchronize()
        self.checks_run += 1
        for a, b in zip(_leaves(beside), _leaves(self.outputs[i])):
            if not th.equal(a, b):
                bad = int((a != b).sum())
                raise RuntimeError(
                    f"GraphReplicas: graph {i} produced {bad} different values next to the other "
                    f"stream(s) than alone on the chip (check {self.checks_run}, submission "
                    f"{self._submitted}): kernels of different streams disturb each other on this "
                    "build / driver -- run with replicas=1")

    def wait(self, index: int) -> Any:
        self.streams[index % self.replicas].synchronize()
        return self.outputs[index]

    def synchronize(self) -> None:
        for stream in self.streams:
            stream.synchronize()
        if self.streams:  # graph launches cannot copy the status word themselves: read it here
            nn_ops.lstm_timeouts(self.streams[0].device)


def _leaves(o: Any) -> List[th.Tensor]:
    if isinstance(o, th.Tensor):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for item in o for t in _leaves(item)]
    return []


def _clone(o: Any) -> Any:
    if isinstance(o, th.Tensor):
        return o.clone()
    if isinstance(o, (tuple, list)):
        return type(o)(_clone(item) for item in o)
    return o


class PipelinedReplicas:
    """Several batches in flight with the step cut into STAGES at its persistent LSTM launch (round 5):

        stage A  everything up to the mask estimator's LSTM stack   -> the head stream (`front` = "head", the default)
        stage L  the persistent LSTM-stack launch (+ its sentinel fill) -> the head stream: ONE stream for all batches
        stage B  everything behind it (MVDR, features, encoder, head)  -> the batch's worker stream
        (stage B is cut once more where the step calls the hook with "enhance_end" -- EnhASRBase does, behind its front
        end: four graphs per step; a step that never calls it keeps three.  `mid` = "head", the default, replays the part
        in front of that cut (masks, MVDR, features: ~0.13 ms alone) on the head stream, which has the time: 15 230 -
        15 240 against 15 090 - 15 130 utt/s on one box; `mid` = "worker" replays it on the batch's worker, which is
        what a caller that also queues host -> device copies on the head stream wants (bench.py `host_input`: 13.3 k
        against 11.7 k utt/s).  The attribute may be changed between submissions: a graph replays on any stream.)

    Why.  GraphReplicas keeps R whole steps in flight; every one of them contains a persistent LSTM launch whose
    workgroups synchronise through memory, so R of those may meet on the chip, each sized for 1 / R of it
    (`nn_ops.lstm_share`), and R = 3 already loses (10.6 k against 12.6 k utt/s).  Here the LSTM launches of ALL
    batches run one after the other on their own stream -- never two at once -- while `workers` other streams run
    the GEMM-bound rest of `workers` batches beside the one LSTM: the launches of the 32-utterance step leave the chip
    half empty (252 - 756 four-wave tiles on 256 CUs, ~5 us of launch floor each), which is what more batches in
    flight fill.  Measured (joint step of BASELINE configs[4], 32 utterances, same box, GPU_MAX_HW_QUEUES=8;
    profiles/r05_pipeline_sweep.txt): whole steps on 2 streams 12 240 utt/s (2.62 ms), 3 workers + the LSTM stream
    14 230 (2.25 ms) with the LSTM sized for half the chip (`lstm_share` = 2: it leaves the GEMMs of three batches
    more of every CU than a full-size launch would: 10 570 with `lstm_share` = 1), 4 workers 12 050, 5 workers 10 950,
    6 workers 14 770 against 15 530 for 3 on a later box: FOUR busy streams are what the chip runs without loss -- a
    fifth busy queue shares a dispatch pipe with another, the pair stops overlapping, and with batches dealt
    round-robin the slowest stream sets the pace.  Under load (scripts/pipeline_stage_times.py): stage A 0.38 ms, the
    LSTM 1.28 (alone 1.0), stage B 5.9 (alone 2.3); the workers are busy 0.97 of the time, the head stream 0.82.
    Stage A (STFT, features, the LSTM's input projection: ~0.25 ms) runs on the head stream in front of its batch's LSTM
    launch: on the batch's worker stream (`front` = "worker", the first form) that stream sits idle while the batch
    waits for its turn at the LSTM (in-order streams), 14 510 against 15 130 utt/s on one box; on a stream of its own
    (a fifth active stream) 11 040 -- as with four workers, more than four busy streams lose.
    It needs its streams on their OWN hardware queues: HIP multiplexes streams onto 4 by default, so the runtime has
    to be started with GPU_MAX_HW_QUEUES >= workers + 2 (11 280 utt/s on 4 queues against 14 670 on 8).

    Each stage of each resident batch is its own hipGraph; the stages of one batch share a graph memory pool and are
    captured from ONE call of the step function: `nn_ops.STAGE_HOOK` ends the capture on the worker stream and opens
    the next one on the LSTM stream (and back) at the launch.  A submission replays A on the worker, L on the LSTM
    stream behind an event, B on the worker behind another: three graph launches and two event pairs per step.
    A step without such a launch is a single stage (then this class is GraphReplicas with `workers` streams).

    fn: a list of no-argument step callables, one per resident batch (as for GraphReplicas); lstm_share: what the
    persistent launches are sized for while this object lives (held like GraphReplicas holds its share); verify:
    every pipeline reproduces the eager step bit for bit right after capture, replay after replay.
    """

    def __init__(self, fns, workers: int = 3, lstm_share: int = 2, verify: bool = True, front: str = "head",
                 mid: str = "head", lookahead: bool = False, heads: int = 1) -> None:
        if workers < 1 or lstm_share < 1:
            raise ValueError(f"workers and lstm_share must be >= 1, got {workers}, {lstm_share}")
        if front not in ("head", "worker"):
            raise ValueError(f"front must be head | worker, got {front}")
        if mid not in ("head", "worker"):
            raise ValueError(f"mid must be head | worker, got {mid}")
        self.front, self.mid = front, mid
        # lookahead (round 6): submit() launches the front of batch k (stage A + its persistent LSTM launch) and the back
        # of batch k - workers (the front end's tail + the encoder).  With stage A on the batch's WORKER stream the head
        # stream then carries nothing but the LSTM launches, back to back, and a worker's next stage A is queued in FRONT
        # of its current batch's back stages -- it never idles behind an LSTM wait (the reason round 5 kept stage A on
        # the head stream, which made the head stream's A + L + M = 2.0 ms per step the bound of the pipeline).
        self.lookahead = bool(lookahead)
        _native.load()
        self.fns = list(fns)
        if self.lookahead and len(self.fns) < 2 * workers:
            raise ValueError(f"lookahead needs at least 2 x workers = {2 * workers} resident batches")
        self.workers = workers
        self.lstm_share = lstm_share
        dev = th.device("cuda", th.cuda.current_device())
        queues = hardware_queues()
        if queues < workers + 2:
            import warnings
            warnings.warn(f"PipelinedReplicas: {workers} worker streams + the LSTM stream + the caller's stream need "
                          f"{workers + 2} hardware queues, the HIP runtime was started with {queues} (streams beyond that "
                          "share a queue, i.e. run one after the other: measured 11.3 k against 14.7 k utt/s). Set "
                          "GPU_MAX_HW_QUEUES=8 in the environment BEFORE the process touches the GPU (bench.py does).")
        self.checks_run = 0
        self._saved_in_flight = nn_ops.STREAMS_IN_FLIGHT
        nn_ops.STREAMS_IN_FLIGHT = workers + 1
        nn_ops.push_lstm_share(lstm_share)
        self._open = True
        try:
            distinct = list(dict.fromkeys(self.fns))
            eager = {f: _clone(f()) for f in distinct}       # (the persistent launches sized as nn_ops.lstm_share() says)
            self.eager_outputs = [eager[f] for f in self.fns]
            th.cuda.synchronize()
            # heads > 1 (experiment, round 6): the persistent launches of consecutive batches alternate over `heads`
            # streams -- that many of them may be on the chip at once, so `lstm_share` has to cover them
            self.heads = int(heads)
            streams = replica_streams(dev, workers + self.heads)
            self.streams, self.lstm_stream = streams[:workers], streams[workers]
            self.lstm_streams = streams[workers:]
            self.front_stream = self.lstm_stream if front == "head" else None
            self.pipelines: List[List[Tuple[th.cuda.CUDAGraph, bool]]] = []   # per batch: (graph, the LSTM stage?)
            self.kinds: List[List[str]] = []   # per batch and stage: "a" front | "l" LSTM | "m" up to `enhance_end` | "b" rest
            self.outputs: List[Any] = []
            for i, f in enumerate(self.fns):
                segs, kinds, out = self._capture(f, self.streams[i % workers])
                self.pipelines.append(segs)
                self.kinds.append(kinds)
                self.outputs.append(out)
        except BaseException:
            self.close()
            raise
        self._next = 0
        self._done = [None] * len(self.pipelines)
        self._timed, self._begin, self._pending = {}, {}, []
        self.stage_log = None   # set to a list: every stage's (kind, start event, end event) is appended
        if verify:
            for rnd in range(3):
                for _ in range(2 * len(self.pipelines)):
                    self.submit()
                self.synchronize()
                self.check_outputs(self.eager_outputs, f"after {rnd + 1} rounds of replays")

    def _capture(self, f, worker: th.cuda.Stream):
        pool = th.cuda.graph_pool_handle()
        segs: List[Tuple[th.cuda.CUDAGraph, bool]] = []
        kinds: List[str] = []
        state = {"graph": None, "kind": "a"}

        def begin(stream: th.cuda.Stream, kind: str) -> None:
            th.cuda.set_stream(stream)
            g = th.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            state["graph"], state["kind"] = g, kind

        def end() -> None:
            state["graph"].capture_end()
            segs.append((state["graph"], state["kind"] == "l"))
            kinds.append(state["kind"])
            state["graph"] = None

        def hook(what: str) -> None:
            if what == "lstm_begin":
                end()
                begin(self.lstm_stream, "l")
            elif what == "lstm_end":
                end()
                begin(worker, "m")
            elif what == "enhance_end" and state["kind"] == "m":
                end()   # the front end's tail (masks, MVDR, features) may stay on the head stream (`mid`), the encoder goes to the worker
                begin(worker, "b")

        before = th.cuda.current_stream()
        th.cuda.synchronize()
        nn_ops.STAGE_HOOK = hook
        try:
            begin(worker, "a")
            out = f()
            end()
            if kinds[-1] == "m":   # no `enhance_end` behind the LSTM: everything behind it is the worker's
                kinds[-1] = "b"
        finally:
            nn_ops.STAGE_HOOK = None
            th.cuda.set_stream(before)
            if state["graph"] is not None:   # (a failure inside a capture: close it so the stream is usable again)
                try:
                    state["graph"].capture_end()
                except Exception:  # noqa: BLE001
                    pass
        return segs, kinds, out

    def __len__(self) -> int:
        return len(self.pipelines)

    @property
    def stages(self) -> int:
        return len(self.pipelines[0]) if self.pipelines else 0

    @property
    def next_index(self) -> int:
        """the resident batch the next submit() launches"""
        return self._next

    def done_event(self, index: int):
        """the event behind the last stage of batch `index`'s latest submission (None before its first): a caller that
        refills the batch's input buffers queues the copy behind it (`stream.wait_event`)"""
        return self._done[index]

    def _run_stages(self, i: int, k0: int, k1: int, prev, timed: bool):
        """stages k0 .. k1 - 1 of batch i, each behind `prev` (an event or None) on its stream; returns the last event"""
        worker = self.streams[i % self.workers]
        staged = len(self.pipelines[i]) > 1
        last = len(self.pipelines[i]) - 1
        for k in range(k0, k1):
            graph = self.pipelines[i][k][0]
            kind = self.kinds[i][k]
            head = self.lstm_streams[i % self.heads]
            st = head if kind == "l" or (kind == "m" and self.mid == "head") else worker
            if k == 0 and staged:
                # the batch's previous pass (its last stage ran on the worker) has to be through with the batch's
                # buffers; `front` = "head": stage A of every batch on the head stream (a worker never idles behind an
                # LSTM wait)
                if self.front_stream is not None:
                    st = self.front_stream
                if self._done[i] is not None and st is not worker:
                    st.wait_event(self._done[i])
            if prev is not None:
                st.wait_event(prev)
            with th.cuda.stream(st):
                if timed and k == 0:
                    self._begin[i] = th.cuda.Event(enable_timing=True)
                    self._begin[i].record(st)
                if self.stage_log is not None:   # (measurements: an event pair around every stage)
                    e0 = th.cuda.Event(enable_timing=True)
                    e0.record(st)
                graph.replay()
                prev = th.cuda.Event(enable_timing=(timed and k == last) or self.stage_log is not None)
                prev.record(st)
                if self.stage_log is not None:
                    self.stage_log.append((kind, e0, prev))
        return prev

    def submit(self, after_caller: bool = True, timed: bool = False) -> Tuple[int, Any]:
        """launch the next batch's stages; its outputs are valid once its worker stream has been waited on
        (wait(index) / synchronize()).  after_caller: as in GraphReplicas.submit -- the batch's first stage is ordered
        behind the work already queued on the caller's stream, so inputs written there are visible (its later stages
        follow the first by events); callers whose inputs do not change between submissions pass False.
        timed: bracket the batch with timing events (first stage begins -> last stage ends on the GPU);
        `latency_ms(index)` reads them once the batch is through.
        With `lookahead` the call launches the FRONT of the next batch (up to and including its first persistent
        launch) and the BACK of the batch submitted `workers` calls earlier -- whose index and outputs it returns
        (None, None while the pipeline fills); flush() / wait() / synchronize() launch the backs still pending."""
        i = self._next
        self._next = (i + 1) % len(self.pipelines)
        prev = None
        if after_caller:
            prev = th.cuda.Event()
            prev.record(th.cuda.current_stream())
        n = len(self.pipelines[i])
        if not self.lookahead or n == 1:
            prev = self._run_stages(i, 0, n, prev, timed)
            self._done[i] = prev
            if timed:
                self._timed[i] = (self._begin[i], prev)
            return i, self.outputs[i]
        cut = self.kinds[i].index("l") + 1
        self._pending.append((i, self._run_stages(i, 0, cut, prev, timed), timed))
        if len(self._pending) > self.workers:
            return self._finish_one()
        return None, None

    def _finish_one(self) -> Tuple[int, Any]:
        i, ev, timed = self._pending.pop(0)
        prev = self._run_stages(i, self.kinds[i].index("l") + 1, len(self.pipelines[i]), ev, timed)
        self._done[i] = prev
        if timed:
            self._timed[i] = (self._begin[i], prev)
        return i, self.outputs[i]

    def flush(self) -> None:
        """launch the back stages of every batch whose front is in flight (lookahead mode; a no-op otherwise)"""
        while self._pending:
            self._finish_one()

    def latency_ms(self, index: int) -> float:
        """GPU time from the start of batch `index`'s first stage to the end of its last stage, of its latest
        submit(timed=True) (waits for the batch)"""
        begin, end = self._timed[index]
        end.synchronize()
        return begin.elapsed_time(end)

    def wait(self, index: int) -> Any:
        if any(i == index for i, _, _ in self._pending):
            self.flush()
        self.streams[index % self.workers].synchronize()
        return self.outputs[index]

    def synchronize(self) -> None:
        self.flush()
        for st in self.streams + list(self.lstm_streams):
            st.synchronize()
        nn_ops.lstm_timeouts(self.streams[0].device)

    def check_outputs(self, want: List[Any], what: str = "") -> None:
        """every resident batch's outputs against `want` (the eager outputs), bit for bit; `checks_run` counts the calls"""
        self.checks_run += 1
        for i, out in enumerate(self.outputs):
            for a, b in zip(_leaves(out), _leaves(want[i])):
                if not th.equal(a, b):
                    raise RuntimeError(f"PipelinedReplicas: batch {i} differs from the eager step {what} "
                                       f"({int((a != b).sum())} values)")

    def close(self) -> None:
        """give the library state back (idempotent; also called on collection)"""
        if self._open:
            self._open = False
            nn_ops.STREAMS_IN_FLIGHT = self._saved_in_flight
            nn_ops.pop_lstm_share(self.lstm_share)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def hardware_queues() -> int:
    """hardware queues the HIP runtime multiplexes its streams onto: GPU_MAX_HW_QUEUES as the process started with
    it (read when the runtime initialises; 4 when unset)"""
    import os
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4

"""GuidedPathTracer — host-side mirror of the reference integrator's interface for this path.

Same property names / defaults / semantics as guided_path.cpp:1014-1085; `render()` follows
GuidedPathTracer::render → renderSPP / renderTime (guided_path.cpp:1342-1585) step by step through the
C-ABI's phase calls, so that a `reducer` can all-reduce the SD-tree statistics between the render passes
and the build of every iteration (multi-GPU, see distributed.py).  With reducer=None the result is
identical to the single call ppg_render().
"""
import math
import time

from .bindings import Engine


class GuidedPathTracer:
    def __init__(self, engine=None, reducer=None, log=None, **props):
        self.engine = engine if engine is not None else Engine.hip(**props)
        self.props = self.engine.props
        self.reducer = reducer
        self.log = log
        self.iterations = []  # one dict per iteration: passes, pass stats, tree stats

    # performRenderPasses (guided_path.cpp:1210-1329) with the optional cross-rank reduction in the middle
    def _passes(self, n):
        e = self.engine
        if self.reducer is None:
            return e.render_passes(n)
        # A rank that is cancelled or fails here still enters the exchange the others are in — with its status word set (distributed._Status),
        # which travels with the data: all ranks see the same sum and leave at the same point.  WHICH exchange that is follows from what every
        # rank knows (final flag, budget type, pass count), never from this rank's own outcome.
        failure = None
        try:
            e.render_passes_nostat(n)
        except Exception as ex:  # PPGError: cancelled, round hook failed, ...
            failure = ex
            self.reducer.status = 1
        try:
            if self._final and self.props["budgetType"] == "spp" and self.reducer.world > 1:
                # a final iteration: the groups of passes were dealt to the ranks — whole, or by tiles —, their partial images add up in group order
                expect = e.final_partials_expected(n)
                ptr, count = None, 0
                if failure is None:
                    try:
                        ptr, count = e.final_partials()
                    except Exception as ex:
                        failure = ex
                if count != expect:
                    self.reducer.status, ptr = 1, None
                self.reducer.reduce_final_partials(e, ptr, expect)
                self._film_complete = True
            else:
                self.reducer.reduce_images(e)
        except Exception:
            if failure is None:
                raise
        if failure is not None:
            raise failure
        return e.finish_passes()

    def _build(self):
        if self.reducer is not None and self._recorded:  # (an iteration that was final from its first pass recorded nothing)
            self.reducer.reduce_sdtree(self.engine)
        return self.engine.build_sdtree()

    def cancel(self):
        """Integrator::cancel (GP:1643-1648), from any thread.  Sharded: the other ranks learn of it in the next exchange and leave too."""
        self._cancelled = True
        self.engine.cancel()

    def _do_nee(self, spp):
        nee = self.props["nee"]
        return False if nee == "never" else (spp < 128 if nee == "kickstart" else True)

    def render(self, scene=None):
        e, p = self.engine, self.props
        if scene is not None:
            e.set_scene(scene)
        # (cancel() is sticky in the library: one that arrived before this call cancels this render in begin_render below; the flag the hooks
        # read is cleared BEFORE that, so that a cancel() arriving any time after this line is seen by them)
        self._cancelled = False
        if self.reducer is not None:
            self.reducer.begin_render()
        try:
            e.begin_render()
        except Exception:
            if self.reducer is None:
                raise
            # Sharded: this rank cannot start (a cancel that arrived before the call, an error) — the peers would wait for it in their first
            # collective.  Every sharded render therefore starts with ONE status exchange; all ranks leave together.
            self.reducer.status = 1
            self.reducer.stop_decision(0)
            raise
        if self.reducer is not None and self.reducer.stop_decision(0):
            from .distributed import RenderAborted
            raise RenderAborted("render aborted: a rank could not start (cancelled or failed before its first exchange)")
        # Sharded with a time budget: every control decision of renderTime() (guided_path.cpp:1434-1514) and the per-pass abort inside
        # performRenderPasses (GP:1259-1262) reads a clock — rank 0's, broadcast, so that all ranks render the same passes and iterations
        clock = (lambda v: self.reducer.broadcast(v)) if self.reducer is not None else (lambda v: v)
        def stop_hook(local):  # rank 0's decision for all — or "stop" when any rank's status word is set (a cancelled rank meets the others here)
            if self._cancelled or local == 2:  # (2 = PPG_STOP_CANCELLED: the library's render was cancelled, by whatever route)
                self.reducer.status = 1
            return self.reducer.stop_decision(local)
        # (hooks of an earlier render() of this engine with another reducer / budget must not survive it)
        e.set_stop_hook(stop_hook if (self.reducer is not None and p["budgetType"] != "spp") else None)
        e.set_pass_hook(None)
        if self.reducer is not None and p["bsdfSamplingFractionLoss"] != "none":
            def round_hook():  # per round: records to the owners of their D-trees, the owners' state back to all
                if self._cancelled:  # (the library keeps a cancelled rank in step with the others' round hooks, with empty rounds)
                    self.reducer.status = 1
                self.reducer.reduce_adam(e)
            e.set_pass_hook(round_hook)
        self.iterations = []
        self._film_complete = False  # the last iteration's film was completed by the exchange of a final iteration's groups
        spp = p["sppPerPass"]
        automatic = p["sampleCombination"] == "automatic"
        it, passes_rendered = 0, 0
        current_var_at_end = float("inf")
        t_start = time.monotonic()
        if p["budgetType"] == "spp":  # renderSPP, guided_path.cpp:1342-1426
            n_passes = int(math.ceil(int(p["budget"]) / float(spp)))
            while passes_rendered < n_passes:
                spp_rendered = passes_rendered * spp
                e.set_do_nee(self._do_nee(spp_rendered))
                remaining = n_passes - passes_rendered
                this_iter = min(remaining, 1 << it)
                if remaining - this_iter < 2 * this_iter:
                    this_iter = remaining
                self._final = this_iter >= remaining
                self._recorded = not self._final  # training passes of this iteration went into the building tree (also when FINAL passes follow, GP:1400-1411)
                self._film_complete = False
                e.begin_iteration(self._final)
                st = self._passes(this_iter)
                rec = dict(iter=it, passes=this_iter, stats=[st.as_dict()])
                passes_rendered += st.passes_rendered_local
                last_var_at_end = current_var_at_end
                current_var_at_end = this_iter * st.variance / remaining
                remaining -= this_iter
                if automatic and remaining > 0 and (remaining < this_iter or (spp_rendered > 256 and current_var_at_end > last_var_at_end)):
                    self._final = True
                    e.set_final(True)
                    st = self._passes(remaining)
                    rec["final_passes"] = remaining
                    rec["stats"].append(st.as_dict())
                    passes_rendered += st.passes_rendered_local
                rec["tree"] = self._build().as_dict()
                e.end_iteration()
                self.iterations.append(rec)
                if self.log:
                    self.log(rec)
                it += 1
        else:  # renderTime, guided_path.cpp:1434-1514
            n_seconds = float(p["budget"])
            elapsed = 0.0
            while elapsed < n_seconds:
                spp_rendered = passes_rendered * spp
                e.set_do_nee(self._do_nee(spp_rendered))
                remaining_time = n_seconds - elapsed
                this_iter = 1 << it
                t_iter = time.monotonic()
                self._final = False
                self._recorded = True
                e.begin_iteration(False)
                st = self._passes(this_iter)
                rec = dict(iter=it, passes=this_iter, stats=[st.as_dict()])
                passes_rendered += st.passes_rendered_local
                seconds_iter = clock(time.monotonic() - t_iter)
                last_var_at_end = current_var_at_end
                current_var_at_end = seconds_iter * st.variance / remaining_time
                remaining_time -= seconds_iter
                if automatic and remaining_time > 0 and (remaining_time < seconds_iter or (spp_rendered > 256 and current_var_at_end > last_var_at_end)):
                    self._final = True
                    e.set_final(True)
                    while True:
                        st = self._passes(this_iter)
                        rec["stats"].append(st.as_dict())
                        passes_rendered += st.passes_rendered_local
                        elapsed = clock(time.monotonic() - t_start)
                        if elapsed >= n_seconds:
                            break
                rec["tree"] = self._build().as_dict()
                e.end_iteration()
                self.iterations.append(rec)
                if self.log:
                    self.log(rec)
                it += 1
                elapsed = clock(time.monotonic() - t_start)
        if self.reducer is not None and not self._film_complete:
            self.reducer.reduce_film(e, inverse_variance=(p["sampleCombination"] == "inversevar"))
        e.end_render()
        return e.read_film()

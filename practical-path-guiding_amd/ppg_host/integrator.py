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
        e.render_passes_nostat(n)
        self.reducer.reduce_images(e)
        return e.finish_passes()

    def _build(self):
        if self.reducer is not None and not self._final:
            self.reducer.reduce_sdtree(self.engine)
        return self.engine.build_sdtree()

    def _do_nee(self, spp):
        nee = self.props["nee"]
        return False if nee == "never" else (spp < 128 if nee == "kickstart" else True)

    def render(self, scene=None):
        e, p = self.engine, self.props
        if scene is not None:
            e.set_scene(scene)
        e.begin_render()
        if self.reducer is not None and p["budgetType"] != "spp":
            # every control decision of renderTime() (guided_path.cpp:1434-1514) reads a rank-local clock: ranks would render
            # different numbers of passes and iterations and their collectives would no longer match
            raise ValueError("sharded rendering needs budgetType='spp' (a time budget is decided by rank-local clocks)")
        if self.reducer is not None and p["bsdfSamplingFractionLoss"] != "none":
            e.set_pass_hook(lambda: self.reducer.reduce_adam(e))  # per round: records to the owners of their D-trees, the owners' state back to all
        self.iterations = []
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
                e.begin_iteration(False)
                st = self._passes(this_iter)
                rec = dict(iter=it, passes=this_iter, stats=[st.as_dict()])
                passes_rendered += st.passes_rendered_local
                seconds_iter = time.monotonic() - t_iter
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
                        elapsed = time.monotonic() - t_start
                        if elapsed >= n_seconds:
                            break
                rec["tree"] = self._build().as_dict()
                e.end_iteration()
                self.iterations.append(rec)
                if self.log:
                    self.log(rec)
                it += 1
                elapsed = time.monotonic() - t_start
        if self.reducer is not None:
            self.reducer.reduce_film(e, inverse_variance=(p["sampleCombination"] == "inversevar"))
        e.end_render()
        return e.read_film()

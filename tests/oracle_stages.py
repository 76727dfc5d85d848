"""TEST-ONLY stage backend for svgf_amd.strips.StripRunner: the CPU oracle on torch CPU tensors.

The product's strip driver is C++ (svgf_amd/csrc/svgf_strip.hip); StripRunner is a Python restatement of its schedule that runs on any
stage backend.  This one lives under tests/ so that the halo plans and the torch.distributed plumbing can be exercised with gloo on machines
without a GPU; it is never imported by svgf_amd/."""
from __future__ import annotations

from oracle import oracle as orc


class OracleStages:
    def __init__(self, geo, params, storage):
        self.geo, self.p, self.storage = geo, params, storage

    def _geo(self, rows):
        g = self.geo
        return (g.y0, g.y1 - g.y0, rows[0], rows[1])

    @staticmethod
    def gb(d):
        return {k: v.numpy() if hasattr(v, "numpy") else v for k, v in d.items()}

    def temporal(self, rows, prev_colour, radiance, colour_out, gb_cur, gb_prev, hist_prev, hist_cur, mom_cur, mom_prev):
        g, p = self.geo, self.p
        orc.temporal(g.W, g.H, self.storage, prev_colour.numpy(), radiance.numpy(), colour_out.numpy(), self.gb(gb_cur), self.gb(gb_prev),
                     hist_prev.numpy(), hist_cur.numpy(), mom_cur.numpy(), mom_prev.numpy(), depth_threshold=p["depth_threshold"],
                     normal_threshold=p["normal_threshold"], history_base=p["history_base"], mesh_id_test=p["mesh_id_test"],
                     geo=self._geo(rows))

    def moments(self, rows, colour, out, mom, gb, hist):
        g, p = self.geo, self.p
        orc.moments(g.W, g.H, self.storage, colour.numpy(), out.numpy(), mom.numpy(), self.gb(gb), hist.numpy(),
                    phi_colour=p["phi_colour"], phi_normal=p["phi_normal"], radius=p["moments_radius"], geo=self._geo(rows))

    def atrous(self, rows, src, dst, feedback, gb, step, iteration):
        g, p = self.geo, self.p
        orc.atrous(g.W, g.H, self.storage, src.numpy(), dst.numpy(), None if feedback is None else feedback.numpy(), self.gb(gb),
                   step=step, phi_colour=p["phi_colour"], phi_normal=p["phi_normal"], iteration=iteration, geo=self._geo(rows))

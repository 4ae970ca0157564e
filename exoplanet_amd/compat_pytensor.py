"""EXPERIMENTAL (never executed here: PyTensor is not in this image; not imported by the package).
PyTensor Ops over the exoplanet_amd kernels -- the object the reference imports as
``exoplanet.compat.ops`` (``from exoplanet_core.pymc import ops``, compat.py:27,56).

The reference touches exactly three callables on it:

    ops.kepler(M, ecc) -> (sin f, cos f)                    orbits/keplerian.py:333,818
    ops.quad_solution_vector(b, r) -> s[..., 3]             light_curves/limb_dark.py:24
    ops.contact_points(a, e, cosw, sinw, cosi, sini, L)     orbits/keplerian.py:744-753
        -> (M_left, M_right, flag)

This is the Op-level compatibility path: every ``perform`` moves its arrays host -> device ->
host (PyTensor's graph lives on the host), so it is a drop-in, not the fast path -- the fast
path is the fused light curve / GP entry points used by ``exoplanet_amd.LimbDarkLightCurve``
and ``exoplanet_amd.gp``.

PyTensor is not part of this image, so the module is import-guarded and NOT exercised by the
test-suite here (tests/test_host_logic.py only checks the guard); the gradient expressions
restate exoplanet_core's Op gradients in terms of the forward outputs.
"""
import numpy as np

try:  # pragma: no cover - PyTensor is absent from the build image
    import pytensor
    import pytensor.tensor as pt
    from pytensor.graph.basic import Apply
    from pytensor.graph.op import Op
except ImportError as exc:  # pragma: no cover
    raise ImportError(
        "exoplanet_amd.compat_pytensor needs PyTensor (the reference's graph library); "
        "the torch-native API (exoplanet_amd.KeplerianOrbit, LimbDarkLightCurve, gp) does not"
    ) from exc

import torch

from . import ops as _ops


def _dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device="cuda")


class KeplerOp(Op):
    __props__ = ()

    def make_node(self, M, ecc):
        M, ecc = pt.as_tensor_variable(M), pt.as_tensor_variable(ecc)
        return Apply(self, [M, ecc], [M.type(), M.type()])

    def infer_shape(self, fgraph, node, shapes):
        return shapes[0], shapes[0]

    def perform(self, node, inputs, outputs):
        M, ecc = np.broadcast_arrays(*inputs)
        sinf, cosf = _ops.kepler(_dev(M), _dev(ecc))
        outputs[0][0] = sinf.cpu().numpy().reshape(M.shape)
        outputs[1][0] = cosf.cpu().numpy().reshape(M.shape)

    def grad(self, inputs, gradients):
        M, e = inputs
        sinf, cosf = self(M, e)
        ome2 = 1 - e**2
        bM, be = pt.zeros_like(M), pt.zeros_like(M)
        # d(sin f) = cos f df, d(cos f) = -sin f df;  df/dM = (1 + e cos f)^2 / (1 - e^2)^1.5,
        # df/de = (2 + e cos f) sin f / (1 - e^2)
        for g, dfac in ((gradients[0], cosf), (gradients[1], -sinf)):
            if not isinstance(g.type, pytensor.gradient.DisconnectedType):
                bM = bM + g * dfac * (1 + e * cosf) ** 2 / ome2**1.5
                be = be + g * dfac * (2 + e * cosf) * sinf / ome2
        return [bM, be]

    def R_op(self, inputs, eval_points):
        """forward mode (a JVP, not the transposed product `grad` computes): tangents (dM, de) of the inputs ->
        tangents of (sin f, cos f) through df = f_M dM + f_e de"""
        M, e = inputs
        dM, de = eval_points
        if dM is None and de is None:
            return [None, None]
        sinf, cosf = self(M, e)
        ome2 = 1 - e**2
        df = pt.zeros_like(M)
        if dM is not None:
            df = df + dM * (1 + e * cosf) ** 2 / ome2**1.5
        if de is not None:
            df = df + de * (2 + e * cosf) * sinf / ome2
        return [cosf * df, -sinf * df]


class QuadSolutionVectorOp(Op):
    __props__ = ()

    def make_node(self, b, r):
        b, r = pt.as_tensor_variable(b), pt.as_tensor_variable(r)
        out = pt.TensorType("float64", shape=(None,) * (b.ndim + 1))
        return Apply(self, [b, r], [out(), out(), out()])  # s, ds/db, ds/dr

    def infer_shape(self, fgraph, node, shapes):
        shape = tuple(shapes[0]) + (3,)
        return shape, shape, shape

    def perform(self, node, inputs, outputs):
        b, r = np.broadcast_arrays(*inputs)
        s, dsdb, dsdr = _ops.quad_solution_vector_derivs(_dev(b), _dev(r))
        for k, x in enumerate((s, dsdb, dsdr)):
            outputs[k][0] = x.cpu().numpy().reshape(b.shape + (3,))

    def grad(self, inputs, gradients):
        s, dsdb, dsdr = self(*inputs)
        g = gradients[0]
        if isinstance(g.type, pytensor.gradient.DisconnectedType):
            return [pt.zeros_like(inputs[0]), pt.zeros_like(inputs[1])]
        return [pt.sum(g * dsdb, axis=-1), pt.sum(g * dsdr, axis=-1)]


class ContactPointsOp(Op):
    """No gradient: the reference only uses the result to select cadences."""

    __props__ = ()

    def make_node(self, *args):
        args = [pt.as_tensor_variable(a) for a in args]
        return Apply(self, args, [args[0].type(), args[0].type(), pt.TensorType("int32", shape=(None,) * args[0].ndim)()])

    def infer_shape(self, fgraph, node, shapes):
        return shapes[0], shapes[0], shapes[0]

    def perform(self, node, inputs, outputs):
        arrs = np.broadcast_arrays(*inputs)
        Ml, Mr, flag = _ops.contact_points(*[_dev(a) for a in arrs])
        outputs[0][0] = Ml.cpu().numpy().reshape(arrs[0].shape)
        outputs[1][0] = Mr.cpu().numpy().reshape(arrs[0].shape)
        outputs[2][0] = flag.cpu().numpy().astype(np.int32).reshape(arrs[0].shape)


class _Ops:
    """what ``exoplanet/compat.py`` exports as ``ops``"""

    kepler = KeplerOp()
    contact_points = ContactPointsOp()
    _qsv = QuadSolutionVectorOp()

    @classmethod
    def quad_solution_vector(cls, b, r):
        return cls._qsv(b, r)[0]


ops = _Ops()

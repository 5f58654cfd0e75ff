"""HIP-backed replacement of the reference's pybind module `pointnet2._ext` (EXT/src/bindings.cpp:11-24).

Same function names, argument order and error behaviour (RuntimeError for non-contiguous / wrong dtype, the
TORCH_CHECK contract of EXT/include/utils.h:20-45); results are bit-exact with the reference's CPU loops.
Inputs must be HIP device tensors -- there is no CPU path here.

The hot path uses furthest_point_sampling / gather_points / ball_query / group_points.  The backward ops and the
three_nn / three_interpolate family are outside the path (PEM never calls them in inference) and
raise NotImplementedError rather than silently falling back.
"""
from sam6d_hip import ops as _ops


def furthest_point_sampling(points, nsamples):
    return _ops.furthest_point_sampling(points, int(nsamples))


def gather_points(points, idx):
    return _ops.gather_points(points, idx)


def ball_query(new_xyz, xyz, radius, nsample):
    return _ops.ball_query(new_xyz, xyz, float(radius), int(nsample))


def group_points(points, idx):
    return _ops.group_points(points, idx)


def _out_of_path(name):
    def f(*a, **k):
        raise NotImplementedError(
            "pointnet2._ext.%s is outside the inference hot path this library implements "
            "(training/backward and three_nn/three_interpolate are unused by PEM inference)" % name)
    f.__name__ = name
    return f


gather_points_grad = _out_of_path("gather_points_grad")
group_points_grad = _out_of_path("group_points_grad")
three_nn = _out_of_path("three_nn")
three_interpolate = _out_of_path("three_interpolate")
three_interpolate_grad = _out_of_path("three_interpolate_grad")

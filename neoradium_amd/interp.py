"""Tap tables for the non-default interpolators of ``Grid.estimateChannelLsEx`` (reference utils.py:26-35, grid.py:853-861).

Every interpolation kind the reference offers is linear in the sample values for fixed sample and query positions, so
for one pilot geometry it is a sparse (or, for the global quadratic spline, dense) matrix.  This module builds that
matrix on the host as ``(idx, w)`` tap tables -- setup-time work like the RE index maps -- and the per-slot arithmetic is
``nrx_interp_taps_f64`` on the GPU.

    nearest / quadratic   scipy ``interp1d(kind=..., fill_value='extrapolate')`` applied to the identity: the operator of
                          exactly the routine the reference calls (one non-zero per row / a dense row).
    thin_plate_spline,    ``RBFInterpolator`` with ``neighbors`` nearest samples: the published local-RBF construction
    multiquadric, ...     (kernel matrix of the neighbours + a polynomial tail of degree d on coordinates shifted and
                          scaled to the neighbours' bounding box, smoothing on the diagonal), solved here for the weights
                          of the sample values instead of the coefficients; neighbours come from the same KD-tree query so
                          ties are broken the same way.
"""
import itertools

import numpy as np

_SCALE_FREE = ('linear', 'thin_plate_spline', 'cubic', 'quintic')
_MIN_DEGREE = {'multiquadric': 0, 'linear': 0, 'thin_plate_spline': 1, 'cubic': 1, 'quintic': 2}


def _phi(kernel, r):
    if kernel == 'linear':
        return -r
    if kernel == 'thin_plate_spline':
        with np.errstate(divide='ignore', invalid='ignore'):
            v = r * r * np.log(r)
        return np.where(r == 0, 0.0, v)
    if kernel == 'cubic':
        return r ** 3
    if kernel == 'quintic':
        return -r ** 5
    if kernel == 'multiquadric':
        return -np.sqrt(r * r + 1)
    if kernel == 'inverse_multiquadric':
        return 1 / np.sqrt(r * r + 1)
    if kernel == 'inverse_quadratic':
        return 1 / (r * r + 1)
    if kernel == 'gaussian':
        return np.exp(-r * r)
    raise ValueError("`kernel` must be one of linear, thin_plate_spline, cubic, quintic, multiquadric, "
                     "inverse_multiquadric, inverse_quadratic, gaussian.")


def _monomials(ndim, degree):
    rows = []
    for d in range(degree + 1):
        for combo in itertools.combinations_with_replacement(range(ndim), d):
            rows.append([combo.count(v) for v in range(ndim)])
    return np.int64(rows).reshape(-1, ndim)


def rbf_taps(y, x, kernel='thin_plate_spline', neighbors=None, smoothing=0.0, epsilon=None, degree=None, chunk=4096):
    """Weights of RBFInterpolator(y, ., neighbors, smoothing, kernel, epsilon, degree)(x) on the sample values:
    y (n, ndim) sample positions, x (q, ndim) queries -> idx (q, m) int32, w (q, m) float64."""
    from scipy.spatial import KDTree
    y, x = np.float64(y), np.float64(x)
    ny, nd = y.shape
    _phi(kernel, np.zeros(1))
    if epsilon is None:
        if kernel not in _SCALE_FREE:
            raise ValueError("`epsilon` must be specified if `kernel` is not one of %s." % (", ".join(_SCALE_FREE)))
        epsilon = 1.0
    min_degree = _MIN_DEGREE.get(kernel, -1)
    degree = max(min_degree, 0) if degree is None else int(degree)
    if degree < -1:
        raise ValueError("`degree` must be at least -1.")
    m = ny if neighbors is None else min(int(neighbors), ny)
    powers = _monomials(nd, degree)
    if len(powers) > m:
        raise ValueError("At least %d data points are required when `degree` is %d and the number of dimensions is %d." %
                         (len(powers), degree, nd))
    if neighbors is None:
        nbr = np.broadcast_to(np.arange(ny), (len(x), ny))
    else:
        _, nbr = KDTree(y).query(x, m)
        nbr = np.sort(nbr.reshape(len(x), m), axis=1)
    w = np.empty((len(x), m))
    npoly = len(powers)
    for c0 in range(0, len(x), chunk):
        sel = slice(c0, min(c0 + chunk, len(x)))
        yn, xq = y[nbr[sel]], x[sel]                                       # (c, m, nd), (c, nd)
        lo, hi = yn.min(1), yn.max(1)
        shift, scale = (hi + lo) / 2, (hi - lo) / 2
        scale[scale == 0.0] = 1.0
        ye = yn * epsilon
        yh = (yn - shift[:, None]) / scale[:, None]
        lhs = np.zeros((len(xq), m + npoly, m + npoly))
        lhs[:, :m, :m] = _phi(kernel, np.linalg.norm(ye[:, :, None] - ye[:, None, :], axis=-1))
        lhs[:, np.arange(m), np.arange(m)] += smoothing
        pm = np.prod(yh[:, :, None, :] ** powers[None, None], axis=-1)     # (c, m, npoly)
        lhs[:, :m, m:] = pm
        lhs[:, m:, :m] = np.swapaxes(pm, 1, 2)
        vec = np.empty((len(xq), m + npoly))
        vec[:, :m] = _phi(kernel, np.linalg.norm(xq[:, None] * epsilon - ye, axis=-1))
        vec[:, m:] = np.prod(((xq - shift) / scale)[:, None, :] ** powers[None], axis=-1)
        w[sel] = np.linalg.solve(lhs, vec[..., None])[:, :m, 0]            # lhs is symmetric: weights = lhs^-1 vec
    return np.int32(nbr), w


def taps_1d(x, x_new, kind, neighbors=None, smoothing=0.0):
    """Tap table of utils.interpolate(x, ., x_new, kind, neighbors, smoothing) (reference utils.py:26-35) along one axis."""
    x, x_new = np.float64(x), np.float64(x_new)
    n = len(x)
    if kind in ('thin_plate_spline', 'multiquadric'):
        # utils.py:27-28 pass (neighbors, smoothing, kernel, 1) positionally: the 1 lands on epsilon, degree stays default
        return rbf_taps(x[:, None], x_new[:, None], kind, neighbors, smoothing, 1.0, None)
    if kind == 'linear':
        j = np.clip(np.searchsorted(x, x_new, 'left'), 1, n - 1)
        t = (x_new - x[j - 1]) / (x[j] - x[j - 1])
        return np.int32(np.stack([j - 1, j], 1)), np.stack([1 - t, t], 1)
    if kind in ('nearest', 'quadratic'):
        from scipy.interpolate import interp1d
        op = interp1d(x, np.eye(n), kind=kind, axis=0, fill_value='extrapolate')(x_new)        # (n_out, n)
        if kind == 'nearest':
            j = np.argmax(op, axis=1)
            return np.int32(j[:, None]), np.ones((len(x_new), 1))
        return np.int32(np.broadcast_to(np.arange(n), op.shape)), np.ascontiguousarray(op)
    raise ValueError("Unsupported interpolation method '%s'!" % (kind))

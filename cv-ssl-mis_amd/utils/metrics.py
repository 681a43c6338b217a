"""Binary overlap / surface-distance metrics of the validation path, without medpy.

The reference scores a prediction with ``medpy.metric.binary.dc`` and ``medpy.metric.binary.hd95``
(code/val_2D.py:7-15, code/val_3D.py:82-88).  medpy (pinned by the reference's requirements as ``medpy``, 0.4.0
at the time of the survey) is absent from this image; these functions restate its published algorithm on
numpy/scipy:

  dc   = 2 |A & B| / (|A| + |B|)                                   (0.0 when both are empty)
  hd95 = 95th percentile of the union of the two directed surface-distance sets, a surface voxel being an
         object voxel removed by one binary erosion with the 1-connectivity structuring element, and the
         distance of a surface voxel to the other object's surface coming from the Euclidean distance
         transform of the complement of that surface (``voxelspacing`` = EDT sampling).
"""
import numpy as np
from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure


def dc(result, reference):
    result = np.atleast_1d(np.asarray(result).astype(bool))
    reference = np.atleast_1d(np.asarray(reference).astype(bool))
    intersection = np.count_nonzero(result & reference)
    size = np.count_nonzero(result) + np.count_nonzero(reference)
    return 2.0 * intersection / float(size) if size > 0 else 0.0


def _surface_distances(result, reference, voxelspacing=None, connectivity=1):
    result = np.atleast_1d(np.asarray(result).astype(bool))
    reference = np.atleast_1d(np.asarray(reference).astype(bool))
    if not np.any(result):
        raise RuntimeError('The first supplied array does not contain any binary object.')
    if not np.any(reference):
        raise RuntimeError('The second supplied array does not contain any binary object.')
    footprint = generate_binary_structure(result.ndim, connectivity)
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def hd95(result, reference, voxelspacing=None, connectivity=1):
    hd1 = _surface_distances(result, reference, voxelspacing, connectivity)
    hd2 = _surface_distances(reference, result, voxelspacing, connectivity)
    return float(np.percentile(np.hstack((hd1, hd2)), 95))


def hd(result, reference, voxelspacing=None, connectivity=1):
    """Hausdorff distance: the larger of the two directed maximum surface distances (medpy.metric.binary.hd) --
    code/test_CNNVIT.py:37 (which names it ``hd95``)."""
    hd1 = _surface_distances(result, reference, voxelspacing, connectivity).max()
    hd2 = _surface_distances(reference, result, voxelspacing, connectivity).max()
    return float(max(hd1, hd2))


def asd(result, reference, voxelspacing=None, connectivity=1):
    """Average surface distance: mean distance of the surface voxels of ``result`` to the surface of ``reference``
    (medpy.metric.binary.asd; not symmetric) -- code/test_3D_util.py:147-152."""
    return float(_surface_distances(result, reference, voxelspacing, connectivity).mean())


def ravd(result, reference):
    """Relative absolute volume difference as medpy defines it: (|result| - |reference|) / |reference| (signed; the
    reference takes abs() of it, code/test_3D_util.py:149)."""
    result = np.atleast_1d(np.asarray(result).astype(bool))
    reference = np.atleast_1d(np.asarray(reference).astype(bool))
    vol_reference = np.count_nonzero(reference)
    if vol_reference == 0:
        raise RuntimeError('The second supplied array does not contain any binary object.')
    return (np.count_nonzero(result) - vol_reference) / float(vol_reference)

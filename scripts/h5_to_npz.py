"""Convert a reference dataset directory (HDF5 case files) into the ``.npz`` siblings this package reads when h5py is
not installed.

The reference stores every case as ``<root>/data/slices/<case>.h5`` (2-D training slices), ``<root>/data/<case>.h5``
(2-D validation volumes, BraTS / LA 3-D volumes) with two datasets ``image`` and ``label``
(code/dataloaders/acdc_data_processing.py:29-33, dataset.py:61-76, brats2019.py:37-45).  ``dataloaders.dataset.read_case``
opens those files directly when ``import h5py`` works; this image ships no h5py, so run this script once on any machine
that has it:

    python scripts/h5_to_npz.py /path/to/ACDC [--delete-h5]

It walks ``<root>/data`` recursively and writes ``<case>.npz`` (arrays ``image``, ``label``, same dtypes and shapes,
``np.savez_compressed``) next to every ``<case>.h5``.  List files (train_slices.list, val.list, train.txt, ...) are
used unchanged."""
import argparse
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--delete-h5", action="store_true", help="remove each .h5 after its .npz was written and verified")
    args = ap.parse_args()
    try:
        import h5py
    except ImportError:
        sys.exit("this converter needs h5py; run it where the reference's own environment is installed")
    n = 0
    for d, _, files in os.walk(os.path.join(args.root, "data")):
        for f in sorted(files):
            if not f.endswith(".h5"):
                continue
            src, dst = os.path.join(d, f), os.path.join(d, f[:-3] + ".npz")
            with h5py.File(src, "r") as h:
                image, label = h["image"][:], h["label"][:]
            np.savez_compressed(dst, image=image, label=label)
            with np.load(dst) as z:
                assert z["image"].shape == image.shape and z["image"].dtype == image.dtype
                assert np.array_equal(z["label"], label)
            if args.delete_h5:
                os.remove(src)
            n += 1
    print(f"{n} case files converted under {args.root}/data")


if __name__ == "__main__":
    main()

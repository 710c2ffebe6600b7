"""Named-axis tensor container with the API of the reference's `Tensor_class.Tensor`
(/root/reference/TensorNetwork/Tensor_class.py:6-298), written from scratch.

It is a host-side (NumPy) utility: the device library works on plain arrays in canonical layouts
and never sees these objects.  `Network.As`, `Network.TX` and the environment lists hand them out
so that code written against the reference keeps addressing axes by name.

Pickled attributes match the reference (`elem, shape, rank, aggregations, history_axes_names,
axes_names`), so the shipped `trained_*.dat` models load into this class.
"""
import numpy as np


class Tensor:
    """ndarray + one name per axis + a record of fused ("aggregated") axes."""

    def __init__(self, elem=None, shape=None, axes_names=None, scale=1.):
        # Tensor_class.py:39-94: random U[0,1)/scale when only a shape is given
        if elem is None:
            if shape is None:
                raise Exception('You have to provide either the elements of the tensor or its shape')
            elem = np.random.random(size=shape)
            elem /= scale
        self.elem = elem
        self.shape = self.elem.shape
        self.rank = len(self.shape)
        self.aggregations = {}
        self.axes_names = None
        if axes_names is not None:
            try:
                n_names = len(axes_names)
            except TypeError:
                print("=== Warning ===\nThe object that describes the indexes names have at least to support "
                      "the built-in len function.\naxes_names attribute has not been inizialized.")
            else:
                if n_names == self.rank:
                    self.axes_names = np.array(axes_names)
                    self.history_axes_names = [np.array(axes_names)]
                else:
                    print("=== Warning ===\nThe number of names should match the rank of the tensor."
                          "\naxes_names attribute has not been inizialized.")

    # ---- bookkeeping ---------------------------------------------------------------------------
    def update_members(self, axes_names):
        """Refresh names / shape / rank after `elem` changed (Tensor_class.py:244-256)."""
        self.axes_names = np.array(axes_names)
        self.shape = self.elem.shape
        self.rank = len(self.shape)

    def ax_to_index(self, axes):
        """Position(s) of the named axis/axes (Tensor_class.py:219-241)."""
        def one(name):
            return np.where(self.axes_names == name)[0][0]
        if type(axes) == str:
            return one(axes)
        return [one(a) for a in axes]

    def check_names(self):
        print("=" * 10 + "axes_names type" + "=" * 10)
        print(type(self.axes_names))

    def __str__(self):
        print("=" * 10 + " Tensor description " + "=" * 10)
        print("Tensor shape: ", self.shape)
        print("Tensor rank: ", self.rank)
        print("Axes names: ", self.axes_names)
        return ""

    # ---- axis surgery --------------------------------------------------------------------------
    def transpose(self, permutation):
        """Reorder the axes to the given order of names (Tensor_class.py:202-216)."""
        self.elem = np.transpose(self.elem, self.ax_to_index(permutation))
        self.update_members(permutation)

    def aggregate(self, axes_names=None, new_ax_name=None, debug=False):
        """Fuse the named axes, in the given order, into one leading axis (Tensor_class.py:97-159).
        With `axes_names=None` every axis is fused."""
        if new_ax_name is None:
            raise ValueError("You have to provide the name of the new axes")
        if self.axes_names is None:
            raise ValueError("This function can be called only if the axes names are defined")
        if axes_names is None:
            axes_names = self.axes_names
        for name in axes_names:
            assert name in self.axes_names, "The " + name + " axes wasn't found in the tensor"
        fused = self.ax_to_index(axes_names)
        kept = [i for i in range(self.rank) if i not in set(fused)]
        sizes = [self.shape[i] for i in fused]
        self.aggregations[new_ax_name] = dict(zip(axes_names, np.array(sizes)))
        if debug:
            print("Aggregating...", fused + kept)
        kept_shape = [self.shape[i] for i in kept]
        kept_names = [self.axes_names[i] for i in kept]
        self.elem = np.transpose(self.elem, fused + kept).reshape([-1] + kept_shape)
        self.update_members(np.concatenate([[new_ax_name], kept_names]) if kept_names else [new_ax_name])

    def disaggregate(self, ax):
        """Undo `aggregate` for the axis `ax`: its components come first (Tensor_class.py:162-199)."""
        assert ax in self.axes_names, "The " + ax + " ax wasn't found in the tensor."
        assert ax in self.aggregations.keys(), "The " + ax + " does not represent an aggregated ax."
        parts = self.aggregations[ax]
        names, sizes = list(parts.keys()), [int(v) for v in parts.values()]
        pos = self.ax_to_index(ax)
        rest_names = [n for i, n in enumerate(self.axes_names) if i != pos]
        moved = np.moveaxis(self.elem, pos, 0)
        self.elem = moved.reshape(sizes + list(moved.shape[1:]))
        self.update_members(np.concatenate([names, rest_names]) if rest_names else names)
        self.aggregations.pop(ax)

    # ---- arithmetic (operands aligned by axis name; `o` is permuted in place, as the reference) ----
    def _aligned(self, o):
        assert np.all(np.isin(self.axes_names, o.axes_names)), "Error: axes don't match, cannot sum tensors."
        o.transpose(self.axes_names)
        return o.elem

    def __add__(self, o):
        return Tensor(elem=self.elem + self._aligned(o), axes_names=self.axes_names)

    def __sub__(self, o):
        return Tensor(elem=self.elem - self._aligned(o), axes_names=self.axes_names)

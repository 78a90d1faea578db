"""Host-side mirror of the reference's `Sorting` class (MiscFunctions.cs:122-417): the post-process filters that
Game1.Update applies to the finished Color[] (Game1.cs:226-232).  Same names and argument order as the C# methods;
`array` is a (height, width) uint32 picture (or a flat array of width * height), and like the C# methods each call
is ONE in-place pass and returns the array.  The work is done by libsrt.so on the GPU; there is no CPU fallback."""
import numpy as np


class Sorting:
    def __init__(self, renderer):
        self._r = renderer

    def _run(self, method, array, width, height, *args):
        a = np.asarray(array, dtype=np.uint32)
        if a.size != width * height:
            raise ValueError("array has %d pixels, width * height is %d" % (a.size, width * height))
        out = method(a.reshape(height, width), *args)
        return out.reshape(a.shape)

    def BlurArray(self, array, width, height, weight):
        return self._run(self._r.blur_array, array, width, height, weight)

    def ContrastBlurArray(self, array, width, height, weight, contrast):
        return self._run(self._r.contrast_blur_array, array, width, height, weight, contrast)

    def Despeckle(self, array, width, height, weight, threshold):
        return self._run(self._r.despeckle, array, width, height, weight, threshold)

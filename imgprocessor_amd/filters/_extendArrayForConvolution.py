"""``extendArrayForConvolution`` — reference:
imgProcessor/filters/_extendArrayForConvolution.py:5-97.

Returns the padded array ((ky//2 rows, kx//2 columns per side).  'reflect'
repeats the edge pixel (numpy 'symmetric'); modex may be 'wrap'.  The filters
of this package never need the padded copy (borders are resolved while the
tile is staged into LDS); the function exists for callers of the reference
API and runs as a HIP index-remap kernel.
"""
from .. import ops


def extendArrayForConvolution(arr, kernelXY, modex='reflect', modey='reflect', ctx=None):
    if modey != 'reflect':
        # the reference only accepts 'reflect' (and the misspelt 'warp') for modey
        raise Exception('modey not supported')
    if modex not in ('reflect', 'wrap'):
        raise Exception('modex not supported')
    return ops.extend_array(arr, kernelXY, modex, modey, ctx=ctx)

"""Reader for the AIA temperature-response table the reference loads with ``sunpy.io.special.read_genx``
(``sunerf/rendering/density_temperature.py:131``: ``aia_temp_resp.genx``, an IDL ``savegen`` / XDR big-endian file).

Only what the renderer needs is decoded: for every channel structure ``A94 ... A335`` the ``LOGTE float32[101]`` grid
and the ``TRESP float64[101]`` response that follow its second (data-section) name tag.  sunpy is not required.
"""
import numpy as np

CHANNELS = (94, 131, 171, 193, 211, 304, 335)
_N = 101


def read_aia_temp_resp(path: str):
    """-> (logte float32 [7, 101], tresp float64 [7, 101]) in ``CHANNELS`` order."""
    with open(path, 'rb') as fh:
        blob = fh.read()
    logte = np.empty((len(CHANNELS), _N), dtype=np.float32)
    tresp = np.empty((len(CHANNELS), _N), dtype=np.float64)
    for row, channel in enumerate(CHANNELS):
        name = f'A{channel}'.encode()
        name += b'\0' * (-len(name) % 4)                  # XDR strings are padded to 4 bytes
        header_pos = blob.index(name)                     # structure description
        data_pos = blob.index(name, header_pos + len(name))   # data section
        start = data_pos + 32                             # tag + array descriptors precede the numbers
        grid = np.frombuffer(blob, dtype='>f4', count=_N, offset=start)
        resp = np.frombuffer(blob, dtype='>f8', count=_N, offset=start + 4 * _N)
        if not (abs(float(grid[0]) - 4.0) < 1e-6 and abs(float(grid[-1]) - 9.0) < 1e-5 and np.all(np.diff(grid) > 0)
                and np.all(resp >= 0)):
            raise ValueError(f'{path}: unexpected genx layout for channel {channel}')
        logte[row], tresp[row] = grid, resp
    return logte, tresp

"""Device selection with the semantics of the reference's saber/utils/io.py:93-149 (get_available_devices /
determine_device), restricted to what the hot path needs.  There is no CPU execution path for the engine:
selecting a CPU device is reported here and fails when an Engine is created."""
import torch


def determine_device(deviceID: int = 0):
    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        if deviceID >= n:
            print(f"Warning: Requested device {deviceID} but only {n} devices available")
            print("Falling back to device 0")
            deviceID = 0
        return torch.device(f"cuda:{deviceID}")
    print("No ROCm device available (the MI355X engine cannot run on CPU)")
    return torch.device("cpu")


def get_available_devices(deviceID: int = None):
    if deviceID is None:
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return determine_device(deviceID)


# ------------------------------------------------------------------ files either side of the path
def read_micrograph(fname: str):
    """saber/utils/io.py:43-65: -> (data, pixel size in Angstrom or None).  .mrc through saber_amd.utils.mrc, .tif/.tiff through
    saber_amd.utils.tiff (the packages the reference calls are absent here); .dm4 / .ser need hyperspy, as in the reference."""
    if fname.endswith(".mrc"):
        from saber_amd.utils.mrc import read_mrc
        data, vox = read_mrc(fname, permissive=True)
        return data, vox[0]
    if fname.endswith((".tif", ".tiff")):
        from saber_amd.utils.tiff import imread
        return imread(fname), None
    if fname.endswith((".dm4", ".ser")):
        try:
            import hyperspy.api  # noqa: F401
        except ImportError:
            raise ValueError("Hyperspy is not installed. Please install it to read .dm4 or .ser files. (pip install hyperspy)")
        return read_stem_micrograph(fname)
    raise ValueError(f"Unsupported file type: {fname}")


def read_stem_micrograph(input: str):
    """saber/utils/io.py:67-91: hyperspy signal -> (data, pixel size in Angstrom)"""
    import hyperspy.api as hs
    signal = hs.load(input)
    ax = signal.axes_manager[0]
    factor = {"nm": 10, "µm": 1e3, "pm": 1e-3}.get(ax.units)
    if factor is None:
        raise ValueError(f"Unsupported unit: {ax.units}")
    return signal.data, ax.scale * factor


def read_movie(input: str, scale_factor: float):
    """saber/utils/io.py:12-41: a TIFF stack (one file, or a sorted glob of single frames) as float32, Fourier-cropped per frame"""
    import glob

    import numpy as np

    from saber_amd.filters.downsample import FourierRescale2D
    from saber_amd.utils.tiff import imread
    if "*" in input:
        files = sorted(glob.glob(input))
        if not files:
            raise ValueError(f"No files found for pattern: {input}")
        volume = np.stack([imread(f) for f in files])
    else:
        volume = imread(input)
    volume = volume.astype(np.float32)
    if scale_factor > 1:
        for i in range(volume.shape[0]):
            # the reference assigns the smaller frame into the full-size slot, which only broadcasts when the size is unchanged
            # (io.py:37-39); here that case is reported instead of raising numpy's shape error
            small = FourierRescale2D.run(volume[i], scale_factor)
            if small.shape != volume[i].shape:
                raise ValueError(f"read_movie: a {small.shape} frame does not fit the {volume[i].shape} slot (reference io.py:39 fails the same way)")
            volume[i] = small
    return volume


def mask3D_to_tiff(mask3D, output_path: str):
    from saber_amd.utils.tiff import imsave
    imsave(output_path, mask3D)


def save_copick_metadata(config, metadict: dict, output_path: str):
    """saber/utils/io.py:164-180: <overlay_root>/logs/<output_path> as YAML with inline lists.  Needs the `copick` package for the project
    file; `config` may also be an object with .config.overlay_root (what copick.from_file returns)."""
    import os

    import yaml
    if isinstance(config, str):
        import copick
        config = copick.from_file(config)
    overlay_root = config.config.overlay_root
    if overlay_root[:8] == "local://":
        overlay_root = overlay_root[8:]
    basepath = os.path.join(overlay_root, "logs")
    os.makedirs(basepath, exist_ok=True)

    class InlineListDumper(yaml.SafeDumper):
        pass
    InlineListDumper.add_representer(list, lambda d, data: d.represent_sequence("tag:yaml.org,2002:seq", data, flow_style=True))
    with open(os.path.join(basepath, output_path), "w") as f:
        yaml.dump(metadict, f, Dumper=InlineListDumper, default_flow_style=False, sort_keys=False)


def get_metadata(zarr_path: str):
    """saber/utils/io.py:182-196 opens the store and reads attrs['labels'] / attrs['amg'] (the reference function ends there, without a
    return statement); here the two are returned: ({index: class name}, amg parameter dict)."""
    from saber_amd.utils import zarr_v2
    zfile = zarr_v2.open_group(zarr_path, mode="r")
    labels = {i: name for i, name in enumerate(zfile.attrs["labels"])}
    return labels, zfile.attrs["amg"]

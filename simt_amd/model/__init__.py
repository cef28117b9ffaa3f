"""Drop-in mirrors of the reference's `model` package (model/deeplab_multi.py, model/deeplab.py) over the HIP engine.
Put the `simt_amd/` directory on sys.path (tools/_init_paths.py does the same with its hard-coded home directory) and
`from model.deeplab_multi import DeeplabMulti` resolves here."""

"""Drop-in `src` package: the reference's module names (options, moco, contriever, dist_utils, utils)
for the MoCo pre-training path, running on libczc.so (MI355X)."""

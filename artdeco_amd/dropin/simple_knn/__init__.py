"""Drop-in for the `simple_knn` package (Reconstruct/submodules/simple-knn); the extension lives in `simple_knn._C`."""

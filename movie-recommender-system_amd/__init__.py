"""MI355X-native kNN collaborative-filtering engine (hot path of
EloDoyard/movie-recommender-system's shared/predictions.scala) behind a C ABI.

The directory name is not a Python identifier; import it with
``importlib.import_module("movie-recommender-system_amd")``.
"""

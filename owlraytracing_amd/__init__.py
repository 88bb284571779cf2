"""MI355X-native TrueKNN / RT-DBSCAN neighbour-query path behind the OWL C-ABI.

The compute path is the HIP library built from ``owlraytracing_amd/csrc`` (see ``_lib``); this
package is the thin host side above its C-ABI.  Importing the package does not load the library;
the first call that needs it does, and raises if it has not been built.
"""
__version__ = "0.1.0"

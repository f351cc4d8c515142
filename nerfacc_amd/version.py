__version__ = "0.1.0"
# API level of the reference this package mirrors (leejaeyong7/nerfacc, nerfacc/version.py:5)
__reference_version__ = "0.5.3"

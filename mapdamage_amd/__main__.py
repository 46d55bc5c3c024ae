import sys

from .main import entry_point

sys.exit(entry_point())

from json import *  # noqa: F401,F403  (stand-in for simplejson; test infrastructure only)

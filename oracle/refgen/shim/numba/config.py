def reload_config():
    pass

  character(len=*), parameter :: version = 'mom6hip-shim'

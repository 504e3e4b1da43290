/* sa_io.h -- text-format loaders and writers used by the signalMachine drop-in (host only). */
#ifndef SA_IO_H_
#define SA_IO_H_
#include <stdint.h>
#include <stdio.h>
#include "signalalign_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* whole line without the newline, NULL at end of file (heap; caller frees) */
char *sa_read_line(FILE *f);
/* splits on blanks/tabs in place; returns token count, *toks is heap (caller frees the array only) */
int64_t sa_split_ws(char *line, char ***toks);

#ifdef __cplusplus
}
#endif
#endif
